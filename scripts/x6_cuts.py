"""GPU box: scripts/x6_bench.py once per diagnostic build of the library, in fresh processes.  Build them first (CPU container):
  from vistaocr_amd import build; build.build_experiments(out_dir="scripts/_cut<n>", extra_flags=["-DX6_CUT=<n>"])
X6_CUT bits (gemm_x6.hip; WRONG results): 1 no DMA in the loop, 2 no MFMA, 4 no fragment reads, 8 no barrier, 16 every DMA from the first stage (L2
hits), 32 no wait for the DMAs.  usage: python scripts/x6_cuts.py 0 1 16 32 ..."""
import os, sys, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cuts = [int(c) for c in sys.argv[1:]]
code = """
import sys, runpy; sys.path.insert(0, %r)
import vistaocr_amd._lib as L
L.LIB_PATH = %r
sys.argv = ['x6_bench.py']
runpy.run_path(%r, run_name='__main__')
"""
for c in cuts:
    lib = os.path.join(root, "scripts", "_cut%d" % c, "libvocr.so")
    r = subprocess.run([sys.executable, "-c", code % (root, lib, os.path.join(root, "scripts", "x6_bench.py"))], capture_output=True, text=True)
    print("==== variant %d" % c)
    print("\n".join(l for l in r.stdout.splitlines() if "product" in l or "products" in l), r.stderr[-300:] if r.returncode else "")
