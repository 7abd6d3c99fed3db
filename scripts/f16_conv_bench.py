"""GPU box: the fp16-operand conv kernels alone on the chip, per layer shape of the BASELINE stack at batch 32 (configs[4] after
rapid_ds = the configs[1] shapes with 16 input channels in front): the NHWC conversion pass, the all-DMA kernel (conv3x3_h16_kernel),
the register-staged kernel it replaces (conv3x3_f16_kernel) and the fp32 minimal-filtering kernel, forward form.
usage: python scripts/f16_conv_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vistaocr_amd import ops, _lib
from vistaocr_amd._lib import call
dev = torch.device("cuda:0"); lib = _lib.load()
s = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
tot = dict(cvt=0.0, h16=0.0, f16=0.0, f32=0.0)
for (cin, cout, h, w) in [(16, 64, 30, 600), (64, 64, 30, 600), (64, 128, 15, 420), (128, 128, 15, 420), (128, 256, 7, 294), (256, 256, 7, 294), (256, 256, 7, 294)]:
    n = 32
    x = torch.randn(n, cin, h, w, device=dev); wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    pf, pd = ops.conv3x3_pack_f16(wt)
    y = torch.empty(n, cout, h, w, device=dev); x16 = torch.empty(n, cin // 16, h, w, 16, dtype=torch.float16, device=dev)
    t_c = timeit(lambda: call("vocr_f32_to_f16_layouts", x.data_ptr(), x16.data_ptr(), None, n, cin, h, w, s()))
    t_h = timeit(lambda: call("vocr_conv3x3_h16_fwd", x16.data_ptr(), pf.data_ptr(), b.data_ptr(), y.data_ptr(), n, cin, h, w, cout, s()))
    y1 = y.clone()
    t_f = timeit(lambda: call("vocr_conv3x3_f16_fwd", x.data_ptr(), pf.data_ptr(), b.data_ptr(), y.data_ptr(), n, cin, h, w, cout, s()))
    err = float((y - y1).abs().max() / y.abs().max())
    p32 = ops.conv3x3_pack(wt)[0]
    t_w = timeit(lambda: ops.conv3x3_forward(x, p32, b, cout))
    t_wg = t_wl = t_w32 = float("nan")
    dy = torch.randn(n, cout, h, w, device=dev)
    if cin >= 4:
        t_w32 = timeit(lambda: ops.conv3x3_wgrad(x, dy))
    if ops.wgrad_f16_layouts_ok(cin, cout):
        x16p = ops.f16_layouts(x, False, True)[1]
        t_wl = timeit(lambda: ops.f16_layouts(dy, True, True))
        dy16p = ops.f16_layouts(dy, False, True)[1]
        t_wg = timeit(lambda: ops.conv3x3_wgrad(x, dy, f16=True, x16p=x16p, dy16p=dy16p))
    fl = 2.0 * n * h * w * cin * cout * 9
    gb = n * h * w * cin * 6 / 1e9
    print("%3d -> %3d  %2dx%3d : convert %6.1f us (%4.2f TB/s)  h16 %6.1f us = %6.1f TF/s  | staged f16 %6.1f us = %5.1f TF/s | fp32 minimal filtering %6.1f us = %5.1f TF/s | h16 vs f16 max rel diff %.1e"
          % (cin, cout, h, w, t_c, gb / t_c * 1e3, t_h, fl / t_h / 1e6, t_f, fl / t_f / 1e6, t_w, fl / t_w / 1e6, err), flush=True)
    print("      weight gradient: fp16 channel-major kernel %6.1f us = %6.1f TF/s (+ both dy layouts in one pass %5.1f us) | fp32 row-pair kernel %6.1f us = %5.1f TF/s"
          % (t_wg, fl / t_wg / 1e6, t_wl, t_w32, fl / t_w32 / 1e6), flush=True)
    tot["cvt"] += t_c; tot["h16"] += t_h; tot["f16"] += t_f; tot["f32"] += t_w
print("forward stack: convert %.0f + h16 %.0f us | staged f16 %.0f us | fp32 %.0f us" % (tot["cvt"], tot["h16"], tot["f16"], tot["f32"]))
