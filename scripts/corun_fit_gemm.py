"""Probe (GPU box): which workgroups get placed beside the persistent panel GEMM (gemm_dma_kernel: 8 waves x 169 VGPRs, 144 KB LDS)?
Same method as corun_fit.py, the resident kernel is the step's layer-0 weight-gradient pair."""
import ctypes, os, subprocess, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
so = "/tmp/corun_fit.so"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-w", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(root, "scripts", "corun_fit.hip")])
import torch
from vistaocr_amd import ops, _lib
fit = ctypes.CDLL(so)
fit.fit_launch.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2
fit.fit_now.argtypes = [ctypes.c_void_p] * 2
dev = torch.device("cuda:0"); lib = _lib.load()
R, G, D = 9408, 2048, 1024
dg = torch.randn(2, R, G, device=dev) * 0.1; x = torch.randn(R, D, device=dev) * 0.1
dw0 = torch.empty(G, D, device=dev); dw1 = torch.empty(G, D, device=dev)
out = torch.zeros(256, dtype=torch.int64, device=dev); t0 = torch.zeros(1, dtype=torch.int64, device=dev); t1 = torch.zeros(1, dtype=torch.int64, device=dev)
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
def gemm(): ops.gemm_pair(0, 1, 0, G, D, R, dg[0], dg[1], G, x, x, D, dw0, dw1, D)
gemm(); torch.cuda.synchronize()
for vg, lds, thr in ((64, 0, 256), (64, 8192, 256), (64, 12288, 256), (64, 16384, 256), (64, 0, 512), (128, 0, 256), (160, 0, 256), (64, 0, 64)):
    fit.fit_launch(vg, lds, thr, out.data_ptr(), side.cuda_stream); torch.cuda.synchronize()
    out.zero_()
    fit.fit_now(t0.data_ptr(), main.cuda_stream); gemm(); fit.fit_now(t1.data_ptr(), main.cuda_stream)
    time.sleep(0.0002)
    fit.fit_launch(vg, lds, thr, out.data_ptr(), side.cuda_stream)
    torch.cuda.synchronize()
    a, b = int(t0[0]), int(t1[0]); st = sorted((int(v) - a) / 100.0 for v in out.tolist())
    inside = sum(1 for v in st if v < (b - a) / 100.0 - 5)
    print("VGPRs %3d LDS %6d threads %3d: GEMM pair %.0f us; probe workgroups started %3d / 256 before its end; first %.0f us, median %.0f, last %.0f after its start"
          % (vg, lds, thr, (b - a) / 100.0, inside, st[0], st[128], st[-1]), flush=True)
