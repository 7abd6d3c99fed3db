"""Micro-benchmark (GPU box): TFLOP/s of every conv / GEMM shape of the BASELINE step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vistaocr_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
N = 32
print("conv  (cin,cout,h,w)            fwd ms  TF/s | dgrad ms TF/s | wgrad ms TF/s")
for cin, cout, h, w in [(1, 64, 30, 600), (64, 64, 30, 600), (64, 128, 15, 420), (128, 128, 15, 420), (128, 256, 7, 294), (256, 256, 7, 294)]:
    x = torch.randn(N, cin, h, w, device=dev); wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.1; dy = torch.randn(N, cout, h, w, device=dev)
    pf, pd = ops.conv3x3_pack(wt)
    fl = 2.0 * N * h * w * cin * cout * 9
    a = timeit(lambda: ops.conv3x3_forward(x, pf, None, cout)); b = timeit(lambda: ops.conv3x3_forward(dy, pd, None, cin)); c = timeit(lambda: ops.conv3x3_wgrad(x, dy))
    print("  %-28s %6.3f %6.1f | %6.3f %6.1f | %6.3f %6.1f" % ((cin, cout, h, w), a * 1e3, fl / a / 1e12, b * 1e3, fl / b / 1e12, c * 1e3, fl / c / 1e12))
print("gemm (ta,tb,m,n,k)")
M = 294 * 32
shapes = [(0, 1, M, 2048, 128), (0, 1, M, 2048, 1024), (0, 0, M, 1024, 2048), (0, 0, M, 128, 2048), (1, 0, 2048, 1024, M), (1, 0, 2048, 128, M), (1, 0, 2048, 512, M - 32),
          (0, 1, M, 128, 1792), (1, 0, 128, 1792, M), (0, 0, M, 1792, 128), (0, 1, M, 96, 1024), (1, 0, 96, 1024, M), (0, 0, M, 1024, 96)]
for ta, tb, m, n, k in shapes:
    a = torch.randn((k, m) if ta else (m, k), device=dev); b = torch.randn((n, k) if tb else (k, n), device=dev); c = torch.empty(m, n, device=dev)
    t = timeit(lambda: ops.gemm(ta, tb, m, n, k, a, a.shape[1], b, b.shape[1], c, n))
    print("  %-28s %6.3f ms %6.1f TF/s" % ((ta, tb, m, n, k), t * 1e3, 2.0 * m * n * k / t / 1e12))
