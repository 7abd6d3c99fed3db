"""GPU box: a few launches each of the three MFMA kernels that carry the step (conv forward 256->256, its weight gradient,
the LSTM x-projection GEMM) for the rocprofv3 --pmc passes (MFMA-busy, LDS, HBM traffic): profiles/r02*_mfma_busy.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vistaocr_amd import ops
dev = torch.device("cuda:0")
cin, cout, h, w = [int(v) for v in os.environ.get("SHAPE", "256,256,7,294").split(",")]
x = torch.randn(32, cin, h, w, device=dev); wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.1; dy = torch.randn(32, cout, h, w, device=dev)
pf, pd = ops.conv3x3_pack(wt)
for _ in range(5): y = ops.conv3x3_forward(x, pf, None, cout)
for _ in range(5): dw = ops.conv3x3_wgrad(x, dy)
a = torch.randn(9408, 1024, device=dev); b = torch.randn(2048, 1024, device=dev); c = torch.empty(9408, 2048, device=dev)
for _ in range(5): ops.gemm(0, 1, 9408, 2048, 1024, a, 1024, b, 1024, c, 2048)
# both directions of a layer in one launch (vocr_gemm_pair): x-projection NT, data gradient NN (K through both pairs), weight gradient TN
b1 = torch.randn(2048, 1024, device=dev); c1 = torch.empty(9408, 2048, device=dev)
for _ in range(5): ops.gemm_pair(0, 0, 1, 9408, 2048, 1024, a, a, 1024, b, b1, 1024, c, c1, 2048)
dg0 = torch.randn(9408, 2048, device=dev); dg1 = torch.randn(9408, 2048, device=dev); w0 = torch.randn(2048, 1024, device=dev); w1 = torch.randn(2048, 1024, device=dev)
dx = torch.empty(9408, 1024, device=dev)
for _ in range(5): ops.gemm_pair(1, 0, 0, 9408, 1024, 2048, dg0, dg1, 2048, w0, w1, 1024, dx, None, 1024)
dw0 = torch.empty(2048, 1024, device=dev); dw1 = torch.empty(2048, 1024, device=dev)
for _ in range(5): ops.gemm_pair(0, 1, 0, 2048, 1024, 9408, dg0, dg1, 2048, a, a, 1024, dw0, dw1, 1024)
torch.cuda.synchronize()
