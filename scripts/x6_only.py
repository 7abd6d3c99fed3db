"""GPU box: eight x-projection products (vocr_gemm_x6, 9408 x 4096 x 1024) and nothing else, with the library X6LIB names: the program rocprofv3 --pmc
runs in scripts/_x6_clock.sh (clock = GRBM_GUI_ACTIVE / 8 / duration and MFMA-busy cycles of the main launches, printed by scripts/x6_clock.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vistaocr_amd._lib as L
if os.environ.get("X6LIB"): L.LIB_PATH = os.environ["X6LIB"]
import torch
from vistaocr_amd import ops
dev = torch.device("cuda:0")
a = torch.randn(9408, 1024, device=dev); b = torch.randn(2048, 1024, device=dev); b1 = torch.randn(2048, 1024, device=dev)
c = torch.empty(9408, 2048, device=dev); c1 = torch.empty(9408, 2048, device=dev)
xa = ops.x6_planes(a, 9408, 1024, True, 1024); wb = ops.x6_planes(b, 4096, 1024, True, 1024, x2=b1, seg=2048, axis=1)
for _ in range(8): ops.gemm_x6(xa, 9408, 1024, wb, 4096, 1024, 9408, 4096, 1024, c, c1=c1, csplit=2048, ldc=2048)
torch.cuda.synchronize()
