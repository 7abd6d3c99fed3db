"""GPU box: TFLOP/s of the GEMM shapes of the BASELINE step, alone on the chip.  VOCR_GEMM_DMA=0 selects gemm.hip's tile kernel for
every shape (A/B of the panel kernel, gemm_dma.hip); the pair rows are vocr_gemm_pair (both directions of a BiLSTM layer in one launch)."""
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
print("single products (ta, tb, m, n, k, bias)")
for ta, tb, m, n, k, hb in shapes:
    a = torch.randn((k, m) if ta else (m, k), device=dev); b = torch.randn((n, k) if tb else (k, n), device=dev); c = torch.empty(m, n, device=dev)
    bias = torch.randn(n, device=dev) if hb else None
    t = timeit(lambda: ops.gemm(ta, tb, m, n, k, a, a.shape[1], b, b.shape[1], c, n, bias=bias))
    print("  %-32s %7.1f us %6.1f TF/s" % ((ta, tb, m, n, k, hb), t * 1e6, 2.0 * m * n * k / t / 1e12))
print("pairs (mode, ta, tb, m, n, k): both directions of a layer")
pairs = [(0, 0, 1, M, 2048, 1024), (0, 0, 1, M, 2048, 128), (1, 0, 0, M, 1024, 2048), (1, 0, 1, M, 1024, 2048), (1, 0, 0, M, 128, 2048), (0, 1, 0, 2048, 1024, M), (0, 1, 0, 2048, 128, M),
         (0, 1, 0, 2048, 512, M - 32)]
for mode, ta, tb, m, n, k in pairs:
    a0 = torch.randn((k, m) if ta else (m, k), device=dev); a1 = torch.randn_like(a0)
    b0 = torch.randn((n, k) if tb else (k, n), device=dev); b1 = torch.randn_like(b0)
    c0 = torch.empty(m, n, device=dev); c1 = torch.empty(m, n, device=dev) if mode == 0 else None
    t = timeit(lambda: ops.gemm_pair(mode, ta, tb, m, n, k, a0, a1, a0.shape[1], b0, b1, b0.shape[1], c0, c1, n))
    print("  %-32s %7.1f us %6.1f TF/s" % ((mode, ta, tb, m, n, k), t * 1e6, 2 * 2.0 * m * n * k / t / 1e12))
