import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vistaocr_amd import ops
dev = torch.device("cuda:0")
for cin, cout, h, w in [(64, 64, 30, 600), (64, 64, 30, 600), (64, 64, 15, 420), (64, 128, 30, 600), (128, 128, 30, 600), (64, 64, 30, 300)]:
    x = torch.randn(32, cin, h, w, device=dev); dy = torch.randn(32, cout, h, w, device=dev)
    for _ in range(10): ops.conv3x3_wgrad(x, dy)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); ops.conv3x3_wgrad(x, dy); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): ops.conv3x3_wgrad(x, dy)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 30 * 1e6
    print((cin, cout, h, w), "events: min %.1f med %.1f max %.1f us; back-to-back wall %.1f us" % (min(ts), sorted(ts)[10], max(ts), wall))
