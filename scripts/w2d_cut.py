"""GPU box: conv3x3_wgrad_wino2d_kernel with the DMAs of its loop cut (a -DW4_CUT=1 build at scripts/_cut/libvocr.so, wrong results):
what the loop costs when no operand has to arrive."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import vistaocr_amd._lib as L
if len(sys.argv) > 1 and sys.argv[1] == "cut":
    L.LIB_PATH = os.path.join(root, "scripts", "_cut", "libvocr.so")
import torch
from vistaocr_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(40): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for cin, cout, h, w in [(64, 64, 30, 600), (64, 128, 15, 420), (128, 128, 15, 420), (128, 256, 7, 294), (256, 256, 7, 294)]:
    x = torch.randn(32, cin, h, w, device=dev); dy = torch.randn(32, cout, h, w, device=dev)
    c = timeit(lambda: ops.conv3x3_wgrad(x, dy))
    print("  %s wgrad %-22s %7.1f us" % (sys.argv[1] if len(sys.argv) > 1 else "full", (cin, cout, h, w), c * 1e6))
