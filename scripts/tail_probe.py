"""GPU box: how much of the conv/GEMM inefficiency is the last partial round of workgroups?  Same per-WG work,
batch sizes giving 4.375 rounds vs 17.5 rounds of workgroups."""
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
for N in (32, 64, 128, 117):
    for cin, cout, h, w in [(256, 256, 7, 294), (128, 128, 15, 420)]:
        x = torch.randn(N, cin, h, w, device=dev); wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.1
        pf, pd = ops.conv3x3_pack(wt)
        fl = 2.0 * N * h * w * cin * cout * 9
        a = timeit(lambda: ops.conv3x3_forward(x, pf, None, cout))
        segs = N * h * ((w + 31) // 32); wgs = (segs + 3) // 4 * ((cout + 127) // 128)
        print("conv N=%3d %s: %.3f ms %.1f TF/s  WGs %d = %.2f per CU" % (N, (cin, cout, h, w), a * 1e3, fl / a / 1e12, wgs, wgs / 256))
for m in (9408, 9408 * 4, 8192, 16384):
    n, k = 2048, 1024
    a = torch.randn(m, k, device=dev); b = torch.randn(n, k, device=dev); c = torch.empty(m, n, device=dev)
    t = timeit(lambda: ops.gemm(0, 1, m, n, k, a, k, b, k, c, n))
    tiles = ((m + 127) // 128) * (n // 128)
    print("gemm NT m=%d: %.3f ms %.1f TF/s tiles %d = %.2f per CU" % (m, t * 1e3, 2.0 * m * n * k / t / 1e12, tiles, tiles / 256))
