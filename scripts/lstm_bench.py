"""Micro-benchmark (GPU box): per-step time of the LSTM recurrent sweeps at the BASELINE shape (VOCR_LSTM_SWEEP=step for one launch per step, chain8 / chain16 for the older generations)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vistaocr_amd import ops, _lib
from vistaocr_amd._lib import call
T, B, H = 294, 32, int(os.environ.get("H", "512"))
dev = torch.device("cuda:0")
lib = _lib.load()
xproj = torch.randn(2, T * B, 4 * H, device=dev) * 0.1
wf = torch.randn(4 * H, H, device=dev) * 0.05
wr = torch.randn(4 * H, H, device=dev) * 0.05
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
y = torch.empty(T * B, 2 * H, device=dev); gates = torch.empty(2, T * B, 4 * H, device=dev); cell = torch.empty(2, T * B, H, device=dev)
ws = torch.empty(lib.vocr_lstm_workspace_bytes(T, B, H) // 4 + 16, device=dev)
dy = torch.randn(T * B, 2 * H, device=dev) * 0.01
dg = torch.empty(2, T * B, 4 * H, device=dev)
wtf, wtr = ops.transpose2d(wf), ops.transpose2d(wr)
s = torch.cuda.current_stream().cuda_stream
def fwd(): call("vocr_lstm_fwd", xproj.data_ptr(), wf.data_ptr(), wr.data_ptr(), lens.data_ptr(), y.data_ptr(), gates.data_ptr(), cell.data_ptr(), ws.data_ptr(), T, B, H, None, s)
def bwd(): call("vocr_lstm_bwd", dy.data_ptr(), wtf.data_ptr(), wtr.data_ptr(), lens.data_ptr(), gates.data_ptr(), cell.data_ptr(), dg.data_ptr(), ws.data_ptr(), T, B, H, None, s)
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("%s dbg=%s: %.3f ms per sweep, %.2f us per step" % (name, os.environ.get("VOCR_LSTM_DEBUG", "0"), dt * 1e3, dt * 1e6 / T))
