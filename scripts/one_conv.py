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
# round 6: the same three products as bf16x6 (vocr_gemm_x6 on split planes): x-projection (256 x 256 tiles), data gradient (256 x 128), weight gradient (K cut)
g6 = lambda t, rows, k, kc, ld, **kw: ops.x6_planes(t, rows, k, kc, ld, **kw)
xa = g6(a, 9408, 1024, True, 1024); wb = g6(b, 4096, 1024, True, 1024, x2=b1, seg=2048, axis=1)
for _ in range(5): ops.gemm_x6(xa, 9408, 1024, wb, 4096, 1024, 9408, 4096, 1024, c, c1=c1, csplit=2048, ldc=2048)
da = g6(dg0, 9408, 4096, True, 2048, x2=dg1, seg=2048, axis=0); wt6 = g6(w0, 1024, 4096, False, 1024, x2=w1, seg=2048, axis=0)
for _ in range(5): ops.gemm_x6(da, 9408, 4096, wt6, 1024, 4096, 9408, 1024, 4096, dx, ldc=1024)
dgt = g6(dg0, 4096, 9408, False, 2048, x2=dg1, seg=2048, axis=1); xt = g6(a, 1024, 9408, False, 1024)
for _ in range(5): ops.gemm_x6(dgt, 4096, 9408, xt, 1024, 9408, 4096, 1024, 9408, dw0, c1=dw1, rsplit=2048, ldc=1024)
# round 5: the fp16-operand kernels (configs[4]) and one LSTM layer's two sweeps at the bench shape
x_nhwc, x16p = ops.f16_layouts(x, True, True)
dy16p = ops.f16_layouts(dy, False, True)[1]
p16 = ops.conv3x3_pack_f16(wt)[0]
for _ in range(5): y16 = ops.conv3x3_forward_f16(x, p16, None, cout, x_nhwc)
for _ in range(5): dw16 = ops.conv3x3_wgrad(x, dy, f16=True, x16p=x16p, dy16p=dy16p)
from vistaocr_amd import _lib
from vistaocr_amd._lib import call
lib = _lib.load()
T, B, H = 294, 32, 512
xp = (torch.rand(2, T * B, 4 * H, device=dev) - 0.5) * 0.6
wf = (torch.rand(4 * H, H, device=dev) - 0.5) * 0.2; wr = (torch.rand(4 * H, H, device=dev) - 0.5) * 0.2
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
yl = torch.empty(T * B, 2 * H, device=dev); gt = torch.empty(2, T * B, 4 * H, device=dev); cl = torch.empty(2, T * B, H, device=dev)
ws = torch.zeros(lib.vocr_lstm_workspace_bytes(T, B, H) // 4 + 16, device=dev)
dyl = (torch.rand(T * B, 2 * H, device=dev) - 0.5) * 0.02; dgl = torch.empty(2, T * B, 4 * H, device=dev); dbl = torch.empty(2, 4 * H, device=dev)
hw = torch.zeros(4, dtype=torch.int32, device=dev)
wtf, wtr = ops.transpose2d(wf), ops.transpose2d(wr)
st = torch.cuda.current_stream().cuda_stream
for _ in range(4):
    call("vocr_lstm_fwd", xp.data_ptr(), wf.data_ptr(), wr.data_ptr(), lens.data_ptr(), yl.data_ptr(), gt.data_ptr(), cl.data_ptr(), ws.data_ptr(), T, B, H, hw.data_ptr(), st)
    call("vocr_lstm_bwd_bias", dyl.data_ptr(), wtf.data_ptr(), wtr.data_ptr(), lens.data_ptr(), gt.data_ptr(), cl.data_ptr(), dgl.data_ptr(), dbl.data_ptr(), ws.data_ptr(), T, B, H, hw.data_ptr(), st)
torch.cuda.synchronize()
