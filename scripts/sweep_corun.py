"""Probe (GPU box): what a co-running MFMA kernel costs the persistent LSTM sweeps, by the co-runner's MFMA shape
(scripts/mfma_spin.hip: 32x32x2 = 64 cycles of the matrix pipe per instruction, 16x16x4 = 32, 4x4x1 = 8) and waves per SIMD.
The sweeps run at the BASELINE shape (T 294, B 32, H 512); the spinner sits on a lowest-priority stream and outlasts them."""
import ctypes, os, subprocess, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
so = "/tmp/mfma_spin.so"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-w", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(root, "scripts", "mfma_spin.hip")])
import torch
from vistaocr_amd import ops, _lib
from vistaocr_amd._lib import call
spin = ctypes.CDLL(so)
spin.spin_launch.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
T, B, H = 294, 32, 512
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
out = torch.zeros(16, device=dev)
main = torch.cuda.current_stream()
lo, hi = torch.cuda.Stream.priority_range()
side = torch.cuda.Stream(priority=lo)
s = main.cuda_stream
def fwd(): call("vocr_lstm_fwd", xproj.data_ptr(), wf.data_ptr(), wr.data_ptr(), lens.data_ptr(), y.data_ptr(), gates.data_ptr(), cell.data_ptr(), ws.data_ptr(), T, B, H, None, s)
def bwd(): call("vocr_lstm_bwd", dy.data_ptr(), wtf.data_ptr(), wtr.data_ptr(), lens.data_ptr(), gates.data_ptr(), cell.data_ptr(), dg.data_ptr(), ws.data_ptr(), T, B, H, None, s)
CYC = {0: 16 * 64, 1: 32 * 32, 2: 16 * 8, 3: 64 * 4 * 4, 4: 16 * 64 + 256, 5: 32 * 32 + 256, 6: 128 * 8 + 256}      # cycles per spinner iteration (one wave per SIMD)
LABEL = {0: "32x32x2 (64 cyc)", 1: "16x16x4 (32 cyc)", 2: "4x4x1 (8 cyc)", 3: "v_fma only", 4: "32x32x2, 20 % pauses", 5: "16x16x4, 20 % pauses", 6: "4x4x1, 20 % pauses"}
SWEEP_FIRST = os.environ.get("SWEEP_FIRST", "0") == "1"      # 1: the sweep is launched first (its waves are the OLDER ones on every SIMD)
def timed(fn, kind=None, waves=1, ms=2.5):
    res = []
    for _ in range(5):
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        if SWEEP_FIRST:
            ev[2].record(main); fn(); ev[3].record(main)
            time.sleep(0.0002)
        if kind is not None:
            iters = int(ms * 1e-3 * 2.1e9 / CYC[kind] / waves)
            ev[0].record(side)
            spin.spin_launch(kind, 256, 256 * waves, iters, out.data_ptr(), side.cuda_stream)
            ev[1].record(side)
            time.sleep(0.0005)
        if not SWEEP_FIRST:
            ev[2].record(main); fn(); ev[3].record(main)
        torch.cuda.synchronize()
        if kind is None: res.append((ev[2].elapsed_time(ev[3]), 0.0, 0.0))
        else: res.append((ev[2].elapsed_time(ev[3]), ev[0].elapsed_time(ev[1]), ev[0].elapsed_time(ev[3]) if not SWEEP_FIRST else -ev[2].elapsed_time(ev[0])))
    return sorted(res)[len(res) // 2]
for kind in CYC: spin.spin_launch(kind, 256, 256, 10, out.data_ptr(), side.cuda_stream)      # load the code objects
fwd(); bwd(); torch.cuda.synchronize()
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    base = timed(fn)[0]
    print("%s sweep alone  %.3f ms  %.2f us/step" % (name, base, base * 1e3 / T))
    for kind in (4, 0, 3):
        for waves, ms in ((1, 4.0), (2, 4.0)):
            t, sp, end = timed(fn, kind, waves, ms)
            print("%s sweep beside %-20s x %d wave/SIMD: sweep %.3f ms (%.2f us/step, +%.0f %%); spinner ran %.3f ms; sweep ended %.3f ms after the spinner started"
                  % (name, LABEL[kind], waves, t, t * 1e3 / T, (t / base - 1) * 100, sp, end))
