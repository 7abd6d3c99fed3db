"""GPU box: conv weight-gradient kernels alone per BASELINE layer (batch 32), one child process per VOCR_WGRAD_WINO_DMA mode
(the mode is read once per process), with the error against an fp64 reference on a small shape."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from vistaocr_amd import ops
    dev = torch.device("cuda:0")
    def timeit(fn, n=30):
        for _ in range(40): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
    # correctness first: odd width, ragged channels
    for (n, cin, h, w, cout) in [(2, 72, 7, 294, 128), (3, 8, 13, 65, 32), (2, 8, 1, 38, 8), (1, 8, 3, 15, 8), (2, 64, 5, 9, 64), (32, 256, 7, 294, 256)]:
        g = torch.Generator().manual_seed(1)
        x = torch.rand((n, cin, h, w), generator=g) * 2 - 1; dy = torch.rand((n, cout, h, w), generator=g) * 2 - 1
        dw = ops.conv3x3_wgrad(x.to(dev), dy.to(dev)).cpu().double()
        ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 3, 3), dy.double(), padding=1)
        print("  check %-24s max abs err %.3e (ref max %.3e)" % ((n, cin, h, w, cout), float((dw - ref).abs().max()), float(ref.abs().max())))
    tot = 0.0
    for cin, cout, h, w in [(64, 64, 30, 600), (64, 128, 15, 420), (128, 128, 15, 420), (128, 256, 7, 294), (256, 256, 7, 294)]:
        x = torch.randn(32, cin, h, w, device=dev); dy = torch.randn(32, cout, h, w, device=dev)
        fl = 2.0 * 32 * h * w * cin * cout * 9
        c = timeit(lambda: ops.conv3x3_wgrad(x, dy))
        tot += c * (2 if (cin, cout) == (256, 256) else 1)
        print("  wgrad %-22s %7.1f us %6.1f TF algorithmic" % ((cin, cout, h, w), c * 1e6, fl / c / 1e12))
    print("  six launches of a step: %.1f us" % (tot * 1e6))
else:
    for mode in sys.argv[1:] or ["2", "3"]:       # 2: one-row piece stream, 3: row pairs (round 3's modes 0 / 1 are gone)
        print("VOCR_WGRAD_WINO_DMA=%s" % mode, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, VOCR_WGRAD_WINO_DMA=mode))
