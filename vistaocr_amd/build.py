"""Build libvocr.so (HIP kernels + C-ABI) for gfx950 in-tree with hipcc.  No JIT cache: the .so sits next to
the sources so it travels with the repo snapshot to the GPU box.

The build is gated by a HASH, not by file times: csrc/build/libvocr.stamp holds the SHA-256 over every source, header and the
compiler flags ("tree"), hipcc's version string ("compiler") and the library's own SHA-256 ("lib").  A tree whose sources differ
from what the library was built from (a fresh checkout with a shipped .so) rebuilds, and so does a box with another ROCm; an
unchanged tree does not.  On a host WITHOUT hipcc the shipped library is accepted when its sources / flags hash and its own
hash match the stamp (the compiler cannot be compared there), and build() raises only when the sources really differ.
`build_report()` says what happened (the driver's "build exercised" question)."""
import hashlib
import json
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["gemm.hip", "gemm_dma.hip", "gemm_x6.hip", "conv.hip", "conv_wino.hip", "conv_f16.hip", "bn_pool.hip", "lstm.hip", "ctc.hip", "misc.cpp", "comm.cpp"]
HEADERS = [os.path.join(CSRC, "vocr_common.h"), os.path.join(CSRC, "conv_tail.h"), os.path.join(CSRC, "gemm_dma.h"),
           os.path.join(CSRC, "..", "..", "include", "vocr.h")]
LIB = os.path.join(CSRC, "libvocr.so")
STAMP = os.path.join(CSRC, "build", "libvocr.stamp")
ARCH = "gfx950"
FLAGS = ["-O3", "--offload-arch=" + ARCH, "-fPIC", "-std=c++17", "-Wno-pass-failed"]
_REPORT = {"built": False, "reason": "not asked"}


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def _tree_hash():
    """SHA-256 over every source, header and the compiler flags (no compiler: a host without hipcc can still compare it)."""
    h = hashlib.sha256()
    for path in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def tree_hash():
    """Public name of the sources / flags hash: bench.py ties profiles/kernel_traffic.json to the kernels it was measured on."""
    return _tree_hash()


def _compiler_id(hipcc):
    if not hipcc:
        return None
    try:
        return hashlib.sha256(subprocess.run([hipcc, "--version"], capture_output=True, text=True, timeout=60).stdout.encode()).hexdigest()
    except Exception:
        return None


def _stale(want, compiler):
    """None when the shipped library is what `want` (sources + flags) and `compiler` (None: unknown here) would produce."""
    if not os.path.exists(LIB):
        return "no library"
    try:
        st = json.load(open(STAMP))
    except Exception:
        return "no stamp"
    if st.get("tree") != want:
        return "sources or flags changed"
    if compiler is not None and st.get("compiler") is not None and st.get("compiler") != compiler:
        return "compiler changed"
    with open(LIB, "rb") as f:
        if hashlib.sha256(f.read()).hexdigest() != st.get("lib"):
            return "library is not the one the stamp describes"
    return None


def build_report():
    """{'built': bool, 'reason': str} of the last build() call in this process."""
    return dict(_REPORT)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    want, compiler = _tree_hash(), _compiler_id(hipcc)
    why = "forced" if force and hipcc else _stale(want, compiler)
    if why is None:
        _REPORT.update(built=False, reason="stamp matches sources, flags and compiler" if compiler else
                       "no compiler here, shipped binary matches sources and flags")
        return LIB
    if hipcc is None:
        raise RuntimeError("hipcc not found: libvocr.so cannot be built (%s)" % why)
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)

    def cc(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr))
        if verbose and r.stderr:
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(cc, SOURCES))
    r = subprocess.run([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs + ["-ldl"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    with open(LIB, "rb") as f:
        libsum = hashlib.sha256(f.read()).hexdigest()
    json.dump({"tree": want, "compiler": compiler, "lib": libsum}, open(STAMP, "w"))
    _REPORT.update(built=True, reason=why)
    return LIB


def build_experiments(out_dir=None, extra_flags=()):
    """An A/B library with the experiment switches compiled IN (-DVOCR_EXPERIMENTS: the VOCR_* kernel-variant environment variables of
    scripts/ are read again), written to scripts/_cut/libvocr.so - never the library the product loads (scripts/_lib_ab.py points a
    probe at it)."""
    hipcc = _hipcc()
    if hipcc is None:
        raise RuntimeError("hipcc not found")
    out_dir = out_dir or os.path.join(os.path.dirname(CSRC), "..", "scripts", "_cut")
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(out_dir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc] + FLAGS + ["-DVOCR_EXPERIMENTS"] + list(extra_flags) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr))
        objs.append(obj)
    lib = os.path.join(out_dir, "libvocr.so")
    r = subprocess.run([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", lib] + objs + ["-ldl"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    return lib


if __name__ == "__main__":
    if "--experiments" in sys.argv:
        print(build_experiments())
    else:
        print(build(force="--force" in sys.argv, verbose=True), build_report())
