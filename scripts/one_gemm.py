"""GPU box: one GEMM shape alone (for rocprofv3 --kernel-trace: which kernel takes it).  usage: one_gemm.py ta tb m n k [bias]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vistaocr_amd import ops
ta, tb, m, n, k = [int(v) for v in sys.argv[1:6]]
hb = len(sys.argv) > 6 and sys.argv[6] == "1"
dev = torch.device("cuda:0")
a = torch.randn((k, m) if ta else (m, k), device=dev); b = torch.randn((n, k) if tb else (k, n), device=dev); c = torch.empty(m, n, device=dev)
bias = torch.randn(n, device=dev) if hb else None
for _ in range(5): ops.gemm(ta, tb, m, n, k, a, a.shape[1], b, b.shape[1], c, n, bias=bias)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): ops.gemm(ta, tb, m, n, k, a, a.shape[1], b, b.shape[1], c, n, bias=bias)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
ref = (a.t() if ta else a).double() @ (b.t() if tb else b).double() + (bias.double() if hb else 0)
print("%s %.1f us %.1f TF/s  max rel err %.2e" % ((ta, tb, m, n, k), t * 1e6, 2.0 * m * n * k / t / 1e12, float((c.double() - ref).abs().max() / ref.abs().max())))
