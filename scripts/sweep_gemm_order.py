"""Probe (GPU box): a persistent LSTM sweep beside the step's real weight-gradient GEMM pair (dW_ih: 2 x [2048 x 1024 x 9408], TN) on a
lowest-priority side stream, by LAUNCH ORDER: the GEMM first (what the step does: its waves are the older ones on every SIMD) or the
sweep first.  Times of both, and when each ended."""
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
main = torch.cuda.current_stream()
lo, hi = torch.cuda.Stream.priority_range()
side = torch.cuda.Stream(priority=lo)
s = main.cuda_stream
def fwd(): call("vocr_lstm_fwd", xproj.data_ptr(), wf.data_ptr(), wr.data_ptr(), lens.data_ptr(), y.data_ptr(), gates.data_ptr(), cell.data_ptr(), ws.data_ptr(), T, B, H, None, s)
def bwd(): call("vocr_lstm_bwd_bias", dy.data_ptr(), wtf.data_ptr(), wtr.data_ptr(), lens.data_ptr(), gates.data_ptr(), cell.data_ptr(), dg.data_ptr(), dbias.data_ptr(), ws.data_ptr(), T, B, H, None, s)
a0 = torch.randn(9408, 2048, device=dev); a1 = torch.randn(9408, 2048, device=dev); xin = torch.randn(9408, 1024, device=dev)
dw0 = torch.empty(2048, 1024, device=dev); dw1 = torch.empty(2048, 1024, device=dev)
def gemm():
    with torch.cuda.stream(side):
        ops.gemm_pair(0, 1, 0, 2048, 1024, 9408, a0, a1, 2048, xin, xin, 1024, dw0, dw1, 1024)
fwd(); bwd(); gemm(); torch.cuda.synchronize()
def run(fn, order):
    res = []
    for _ in range(7):
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        if order == "alone":
            e[0].record(main); fn(); e[1].record(main); torch.cuda.synchronize(); res.append((e[0].elapsed_time(e[1]), 0.0)); continue
        if order == "gemm_first":
            e[2].record(side); gemm(); e[3].record(side); time.sleep(0.0001)
            e[0].record(main); fn(); e[1].record(main)
        else:
            e[0].record(main); fn(); e[1].record(main); time.sleep(0.0001)
            e[2].record(side); gemm(); e[3].record(side)
        torch.cuda.synchronize()
        res.append((e[0].elapsed_time(e[1]), e[2].elapsed_time(e[3])))
    res.sort()
    return res[len(res) // 2]
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(side); gemm(); e1.record(side); torch.cuda.synchronize()
print("GEMM pair alone %.3f ms" % e0.elapsed_time(e1))
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    for order in ("alone", "gemm_first", "sweep_first"):
        sw, gm = run(fn, order)
        print("%s sweep, %-11s: sweep %.3f ms (%.2f us/step)  GEMM pair %.3f ms" % (name, order, sw, sw * 1e3 / T, gm))
