"""Probe (GPU box): does a workgroup of a given size get placed beside a RUNNING forward sweep?  For each (VGPRs, LDS, threads): the
sweep starts, 150 us later 256 probe workgroups are launched on another stream; reported: how many of them started before the sweep
ended (placed beside it) and when."""
import ctypes, os, subprocess, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
so = "/tmp/corun_fit.so"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-w", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(root, "scripts", "corun_fit.hip")])
import torch
from vistaocr_amd import ops, _lib
from vistaocr_amd._lib import call
fit = ctypes.CDLL(so)
fit.fit_launch.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2
fit.fit_now.argtypes = [ctypes.c_void_p] * 2
T, B, H = 294, 32, 512
dev = torch.device("cuda:0"); lib = _lib.load()
xproj = torch.randn(2, T * B, 4 * H, device=dev) * 0.1
wf = torch.randn(4 * H, H, device=dev) * 0.05; wr = torch.randn(4 * H, H, device=dev) * 0.05
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
y = torch.empty(T * B, 2 * H, device=dev); gates = torch.empty(2, T * B, 4 * H, device=dev); cell = torch.empty(2, T * B, H, device=dev)
ws = torch.empty(lib.vocr_lstm_workspace_bytes(T, B, H) // 4 + 16, device=dev)
out = torch.zeros(256, dtype=torch.int64, device=dev); t0 = torch.zeros(1, dtype=torch.int64, device=dev); t1 = torch.zeros(1, dtype=torch.int64, device=dev)
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
def fwd(): call("vocr_lstm_fwd", xproj.data_ptr(), wf.data_ptr(), wr.data_ptr(), lens.data_ptr(), y.data_ptr(), gates.data_ptr(), cell.data_ptr(), ws.data_ptr(), T, B, H, None, main.cuda_stream)
fwd(); torch.cuda.synchronize()
for vg, lds, thr in ((64, 0, 256), (96, 0, 256), (128, 0, 256), (144, 0, 256), (152, 0, 256), (160, 0, 256), (168, 0, 256), (64, 131072, 256), (128, 131072, 256), (144, 131072, 256),
                     (152, 131072, 256), (160, 131072, 256), (160, 65536, 256), (160, 0, 64), (160, 0, 128), (80, 0, 512), (64, 0, 512)):
    fit.fit_launch(vg, lds, thr, out.data_ptr(), side.cuda_stream); torch.cuda.synchronize()       # load the code object
    out.zero_()
    fit.fit_now(t0.data_ptr(), main.cuda_stream); fwd(); fit.fit_now(t1.data_ptr(), main.cuda_stream)
    time.sleep(0.00015)
    fit.fit_launch(vg, lds, thr, out.data_ptr(), side.cuda_stream)
    torch.cuda.synchronize()
    a, b = int(t0[0]), int(t1[0]); st = sorted((int(v) - a) / 100.0 for v in out.tolist())
    inside = sum(1 for v in st if v < (b - a) / 100.0 - 5)
    print("VGPRs %3d LDS %6d threads %3d: sweep %.0f us; probe workgroups started %3d / 256 before the sweep's end; first %.0f us, median %.0f, last %.0f after the sweep's start"
          % (vg, lds, thr, (b - a) / 100.0, inside, st[0], st[128], st[-1]), flush=True)
