"""GPU box: the BatchNorm backward (vocr_bn_relu_bwd = partial-sum pass + apply pass) and the statistics + apply forward pair alone on the chip at the
step's layer shapes: microseconds and the rate of the bytes it must move (2 reads in the partial-sum pass, 2 reads + 1 write in the apply pass).
Round 6: 136 / 82 / 60 us = 5.4 / 6.3 / 5.6 TB/s alone - what the step pays beyond that (267 us per launch on the 147-MB layers) is the chip shared with
the side stream's weight gradients, not the passes' own efficiency."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vistaocr_amd import _lib
from vistaocr_amd._lib import call
dev = torch.device("cuda:0"); lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr() if t is not None else None
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (n, c, h, w) in [(32, 64, 30, 600), (32, 128, 15, 420), (32, 256, 7, 294)]:
    hw = h * w
    y = torch.randn(n, c, h, w, device=dev); da = torch.randn(n, c, h, w, device=dev); dy = torch.empty_like(y); out = torch.empty_like(y)
    mean = torch.zeros(c, device=dev); invstd = torch.ones(c, device=dev); gamma = torch.ones(c, device=dev); beta = torch.zeros(c, device=dev)
    dgamma = torch.empty(c, device=dev); dbeta = torch.empty(c, device=dev); xs = torch.zeros(c, device=dev)
    ws = torch.empty(lib.vocr_bn_workspace_bytes(n, c, hw) // 4 + 16, device=dev)
    bwd = lambda: call("vocr_bn_relu_bwd", P(da), P(y), P(mean), P(invstd), P(gamma), P(beta), P(xs), P(dy), P(dgamma), P(dbeta), None, n, c, hw, P(ws), s)
    t = timed(bwd)
    mb = y.numel() * 4 / 1e6
    print("N%d C%d %dx%d (%.0f MB per tensor): backward %.1f us = %.2f TB/s over 5 tensor passes" % (n, c, h, w, mb, t, 5 * mb / t))
