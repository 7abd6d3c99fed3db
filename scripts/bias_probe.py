"""GPU box: the backward sweep alone with and without the in-sweep bias-gradient accumulation (BASELINE shape)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vistaocr_amd import ops, _lib
from vistaocr_amd._lib import call
T, B, H = 294, 32, 512
dev = torch.device("cuda:0")
lib = _lib.load()
xproj = torch.randn(2, T * B, 4 * H, device=dev) * 0.1
wf = torch.randn(4 * H, H, device=dev) * 0.05; wr = torch.randn(4 * H, H, device=dev) * 0.05
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
y = torch.empty(T * B, 2 * H, device=dev); gates = torch.empty(2, T * B, 4 * H, device=dev); cell = torch.empty(2, T * B, H, device=dev)
ws = torch.empty(lib.vocr_lstm_workspace_bytes(T, B, H) // 4 + 16, device=dev)
dy = torch.randn(T * B, 2 * H, device=dev) * 0.01
dg = torch.empty(2, T * B, 4 * H, device=dev); dbias = torch.empty(2, 4 * H, device=dev)
wtf, wtr = ops.transpose2d(wf), ops.transpose2d(wr)
s = torch.cuda.current_stream().cuda_stream
call("vocr_lstm_fwd", xproj.data_ptr(), wf.data_ptr(), wr.data_ptr(), lens.data_ptr(), y.data_ptr(), gates.data_ptr(), cell.data_ptr(), ws.data_ptr(), T, B, H, None, s)
def bwd(): call("vocr_lstm_bwd", dy.data_ptr(), wtf.data_ptr(), wtr.data_ptr(), lens.data_ptr(), gates.data_ptr(), cell.data_ptr(), dg.data_ptr(), ws.data_ptr(), T, B, H, None, s)
def bwdb(): call("vocr_lstm_bwd_bias", dy.data_ptr(), wtf.data_ptr(), wtr.data_ptr(), lens.data_ptr(), gates.data_ptr(), cell.data_ptr(), dg.data_ptr(), dbias.data_ptr(), ws.data_ptr(), T, B, H, None, s)
def fwd(): call("vocr_lstm_fwd", xproj.data_ptr(), wf.data_ptr(), wr.data_ptr(), lens.data_ptr(), y.data_ptr(), gates.data_ptr(), cell.data_ptr(), ws.data_ptr(), T, B, H, None, s)
for name, fn in (("fwd", fwd), ("bwd", bwd), ("bwd+bias", bwdb), ("bwd", bwd), ("bwd+bias", bwdb)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("%-9s %.3f ms per sweep, %.2f us per step" % (name, dt * 1e3, dt * 1e6 / T))
ref = dg.sum(1)
print("bias check: max abs diff vs column sums %.3e (scale %.3e)" % (float((dbias - ref).abs().max()), float(ref.abs().max())))
