"""Build libvocr.so (HIP kernels + C-ABI) for gfx950 in-tree with hipcc.  No JIT cache: the .so sits next to
the sources so it travels with the repo snapshot to the GPU box."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["gemm.hip", "gemm_dma.hip", "conv.hip", "conv_wino.hip", "conv_f16.hip", "bn_pool.hip", "lstm.hip", "ctc.hip", "misc.cpp", "comm.cpp"]
LIB = os.path.join(CSRC, "libvocr.so")
ARCH = "gfx950"


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libvocr.so cannot be built")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "vocr_common.h"), os.path.join(CSRC, "conv_tail.h"), os.path.join(CSRC, "gemm_dma.h"),
                                                      os.path.join(CSRC, "..", "..", "include", "vocr.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["-O3", "--offload-arch=" + ARCH, "-fPIC", "-std=c++17", "-Wno-pass-failed"]

    def cc(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc] + flags + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
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
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
