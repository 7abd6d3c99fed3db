import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vistaocr_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
cin, cout, h, w = 256, 256, 7, 294
wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.1
pf, pd = ops.conv3x3_pack(wt)
for N in (1, 2, 3, 4, 7, 8, 11, 32):
    x = torch.randn(N, cin, h, w, device=dev)
    fl = 2.0 * N * h * w * cin * cout * 9
    a = timeit(lambda: ops.conv3x3_forward(x, pf, None, cout))
    print("  pad=%s N=%3d  WGs %5d  %7.1f us  %6.1f TF" % (os.environ.get("VOCR_CONV_LDS_PAD", "0"), N, N * 65, a * 1e6, fl / a / 1e12))
