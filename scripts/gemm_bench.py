"""GPU box: TFLOP/s of the GEMM shapes of the BASELINE step, with and without tail-filling K pieces (VOCR_GEMM_TAILFILL)."""
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
M = 294 * 32
shapes = [(0, 1, M, 2048, 128, 1), (0, 1, M, 2048, 1024, 1), (0, 0, M, 1024, 2048, 0), (0, 0, M, 128, 2048, 0), (1, 0, 2048, 1024, M, 0), (1, 0, 2048, 128, M, 0),
          (1, 0, 2048, 512, M - 32, 0), (0, 1, M, 128, 1792, 1), (1, 0, 128, 1792, M, 0), (0, 0, M, 1792, 128, 0), (0, 1, M, 96, 1024, 1)]
for ta, tb, m, n, k, hb in shapes:
    a = torch.randn((k, m) if ta else (m, k), device=dev); b = torch.randn((n, k) if tb else (k, n), device=dev); c = torch.empty(m, n, device=dev)
    bias = torch.randn(n, device=dev) if hb else None
    t = timeit(lambda: ops.gemm(ta, tb, m, n, k, a, a.shape[1], b, b.shape[1], c, n, bias=bias))
    print("  %-32s %7.1f us %6.1f TF/s" % ((ta, tb, m, n, k, hb), t * 1e6, 2.0 * m * n * k / t / 1e12))
