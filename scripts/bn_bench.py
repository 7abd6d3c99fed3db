"""GPU box: BatchNorm/pool backward passes of the two pooled layers, fused (gather) vs two-pass, alone on the chip."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vistaocr_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for n, cin, c, h, w in [(32, 8, 64, 30, 600), (32, 8, 128, 15, 420), (32, 8, 256, 7, 294), (32, 8, 64, 30, 1178), (32, 8, 64, 30, 1180)]:      # 1178: configs[3]'s widest line (W % 4 = 2)
    oh, ow = h // 2, int(w * 0.7)
    x = torch.randn(n, cin, h, w, device=dev); wt = torch.randn(c, cin, 3, 3, device=dev) * 0.1
    bias = torch.zeros(c, device=dev); gamma = torch.ones(c, device=dev); beta = torch.zeros(c, device=dev)
    u = torch.rand(n, c, 2, device=dev)
    for fused in ("1", "0"):
        os.environ["VOCR_POOL_BWD_FUSED"] = fused
        xs = x.clone().requires_grad_(True)
        out = ops.ConvBnReluFn.apply(xs, wt.clone().requires_grad_(True), bias.clone().requires_grad_(True), gamma.clone().requires_grad_(True),
                                     beta.clone().requires_grad_(True), torch.zeros(c, device=dev), torch.ones(c, device=dev), True, 1e-5, 0.1, False, u, oh, ow)
        dout = torch.randn_like(out)
        t = timeit(lambda: out.backward(dout, retain_graph=True), n=10)
        print("%s pooled layer backward (BN+pool+wgrad+dgrad, tiny cin) fused=%s: %.1f us" % ((n, c, h, w), fused, t * 1e6))
