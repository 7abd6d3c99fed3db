"""GPU box: the panel GEMM's main loop with parts switched off (VOCR_GEMM_DBG bits: 1 no DMA in the loop, 2 no barrier, 4 no MFMA, 8 all
12 DMAs at the head of the step) - results are wrong with bits 1/2/4, only the time matters."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
from vistaocr_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=40):
    for _ in range(60): fn()           # ~40 ms of load first: the first launches of a process run at another clock
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
M = 294 * 32
for mode, ta, tb, m, n, k in [(0, 0, 1, M, 2048, 1024), (1, 0, 0, M, 1024, 2048), (0, 1, 0, 2048, 1024, M)]:
    a0 = torch.randn((k, m) if ta else (m, k), device=dev); a1 = torch.randn_like(a0)
    b0 = torch.randn((n, k) if tb else (k, n), device=dev); b1 = torch.randn_like(b0)
    c0 = torch.empty(m, n, device=dev); c1 = torch.empty(m, n, device=dev) if mode == 0 else None
    t = timeit(lambda: ops.gemm_pair(mode, ta, tb, m, n, k, a0, a1, a0.shape[1], b0, b1, b0.shape[1], c0, c1, n))
    print("  dbg=%%s mode=%%s %%-28s %%7.1f us %%6.1f TF/s" %% (os.environ.get("VOCR_GEMM_DBG", "0"), os.environ.get("VOCR_GEMM_MODE", "0"), (mode, ta, tb, m, n, k), t * 1e6, 2 * 2.0 * m * n * k / t / 1e12), flush=True)
''' % ROOT
for spec in sys.argv[1:] or ["0", "1", "3"]:          # "dbg" or "dbg:mode"
    dbg, _, mode = spec.partition(":")
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VOCR_GEMM_DBG=dbg, VOCR_GEMM_MODE=mode or "0"))
