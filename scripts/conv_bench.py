"""GPU box: conv forward TFLOP/s per BASELINE layer (batch 32), plus a batch sweep on the 256->256 layer that shows the
last-partial-round (tail) effect.  VOCR_CONV_TILE=1/2 forces half/full-size workgroups."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vistaocr_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(40): fn()           # the first launches of a process run at another clock
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
layers = [(1, 64, 30, 600), (64, 64, 30, 600), (64, 128, 15, 420), (128, 128, 15, 420), (128, 256, 7, 294), (256, 256, 7, 294)]
tot_f = tot_t = 0.0
for cin, cout, h, w in layers:
    N = 32
    x = torch.randn(N, cin, h, w, device=dev); wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.1; dy = torch.randn(N, cout, h, w, device=dev)
    pf, pd = ops.conv3x3_pack(wt)
    fl = 2.0 * N * h * w * cin * cout * 9
    a = timeit(lambda: ops.conv3x3_forward(x, pf, None, cout)); b = timeit(lambda: ops.conv3x3_forward(dy, pd, None, cin)); c = timeit(lambda: ops.conv3x3_wgrad(x, dy))
    mult = 2 if (cin, cout) == (256, 256) else 1
    tot_f += fl * mult; tot_t += a * mult
    print("conv %-22s fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF | wgrad %7.1f us %6.1f TF" % ((cin, cout, h, w), a * 1e6, fl / a / 1e12, b * 1e6, fl / b / 1e12, c * 1e6, fl / c / 1e12))
print("forward stack (7 launches, back to back alone): %.1f us -> %.1f TF" % (tot_t * 1e6, tot_f / tot_t / 1e12))
if os.environ.get("SWEEP", "1") == "1":
    cin, cout, h, w = 256, 256, 7, 294
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.1
    pf, pd = ops.conv3x3_pack(wt)
    for N in (8, 11, 16, 21, 22, 24, 32, 33, 44, 48, 64, 96, 128):
        x = torch.randn(N, cin, h, w, device=dev)
        fl = 2.0 * N * h * w * cin * cout * 9
        a = timeit(lambda: ops.conv3x3_forward(x, pf, None, cout))
        print("  N=%3d  %7.1f us  %6.1f TF" % (N, a * 1e6, fl / a / 1e12))
