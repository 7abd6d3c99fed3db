"""GPU: every C-ABI operator against the same op on PyTorch-CPU (the arithmetic the reference delegates to),
on seeded inputs sized so the CPU side finishes in seconds.  Integer outputs bit-exact; fp32 within the stated
tolerance."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from vistaocr_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _close(a, b, rtol, atol, what="", max_bad_frac=0.0):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    if float(bad.double().mean()) > max_bad_frac:
        i = int(torch.nonzero(bad.reshape(-1))[0])
        raise AssertionError("%s: %d/%d mismatches, max abs err %.3e (ref scale %.3e); first @%d got %.6g want %.6g"
                             % (what, int(bad.sum()), a.numel(), float(err.max()), float(b.abs().max()), i,
                                float(a.reshape(-1)[i]), float(b.reshape(-1)[i])))


# widths chosen for the segment geometry (conv.hip): W % 32 = 6 / 4 (294, 420: remainder segments spanning 4 / 5 image rows, H not
# a multiple of that), 1 (11 rows per remainder segment), 15 / 16 (2 rows / 1 row), W < 32 (only remainder segments), H = 1
@pytest.mark.parametrize("n,cin,h,w,cout", [(2, 1, 30, 70, 64), (2, 64, 15, 45, 64), (1, 64, 15, 97, 128), (2, 128, 7, 33, 256),
                                            (1, 16, 9, 40, 64), (1, 3, 12, 31, 16), (2, 72, 7, 294, 128), (1, 64, 15, 420, 64),
                                            (3, 8, 13, 65, 32), (2, 16, 5, 47, 64), (1, 8, 6, 48, 16), (2, 8, 1, 38, 8), (1, 8, 3, 15, 8),
                                            # more than 256 workgroup tiles with a short last round: conv3x3_tail_kernel computes it
                                            # (326 tiles of 128co x 2 segments, the last one half empty; 297 of 64co x 4; 293 with Cout = 72)
                                            (5, 20, 7, 294, 256), (6, 12, 15, 420, 64), (9, 8, 7, 294, 72),
                                            # batch-32 launches: enough tiles for the EIGHT-wave F(4,3) workgroups (>= one per CU), with a cut
                                            # last round (264 tiles of 128co x 4 segments) and without (398)
                                            (32, 16, 7, 294, 256), (32, 8, 15, 420, 128)])
@pytest.mark.parametrize("form", ["transform", "direct"])
def test_conv3x3_fwd_dgrad_wgrad(dev, n, cin, h, w, cout, form, monkeypatch):
    """Both forms of the three conv kernels against F.conv2d: "transform" = minimal filtering along the row (conv_wino.hip, the default
    for Cin >= 4: F(4,3) for forward / data-gradient launches with >= 64 output channels, F(2,3) below, F(3,2) weight gradient),
    "direct" = conv.hip (VOCR_CONV_WINO=0 / VOCR_WGRAD_WINO=0, and what Cin < 4 always uses)."""
    from vistaocr_amd import ops
    monkeypatch.setattr(ops, "_WINO", form == "transform")
    monkeypatch.setattr(ops, "_WINO_WGRAD", form == "transform")
    x = _rand((n, cin, h, w), 1)
    wt = _rand((cout, cin, 3, 3), 2, 0.2)
    bias = _rand((cout,), 3)
    dy = _rand((n, cout, h, w), 4)
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, bias, padding=1)
    yr.backward(dy)
    pf, pd = ops.conv3x3_pack(wt.to(dev))
    y = ops.conv3x3_forward(x.to(dev), pf, bias.to(dev), cout)
    _close(y, yr, 1e-4, 1e-4, "conv fwd")
    dx = ops.conv3x3_forward(dy.to(dev), pd, None, cin)
    _close(dx, xr.grad, 1e-4, 1e-4, "conv dgrad")
    dw = ops.conv3x3_wgrad(x.to(dev), dy.to(dev))
    _close(dw, wr.grad, 1e-4, 2e-4 * math.sqrt(n * h * w), "conv wgrad")
    db = ops.channel_sum(dy.to(dev))
    _close(db, dy.sum((0, 2, 3)), 1e-5, 1e-4, "channel_sum")


# the row-pair weight-gradient kernel's geometry (conv3x3_wgrad_wino2d_kernel): one row and one column (a pair with one real row, a
# one-piece row), two columns / three (pieces with 2 / 3 valid elements, four pairs per 8-slot segment), even and odd heights, rows of
# exactly 7 and 8 slots, channel counts that leave most of a 64 x 64 tile empty, and Cin * Cout % 4 != 0 (falls back to the one-row kernel)
@pytest.mark.parametrize("n,cin,h,w,cout", [(1, 4, 1, 1, 4), (3, 4, 2, 2, 8), (2, 8, 5, 3, 4), (2, 68, 9, 5, 12), (1, 8, 4, 24, 8),
                                            (2, 8, 3, 28, 72), (1, 100, 8, 31, 36), (2, 5, 6, 17, 7), (4, 6, 7, 9, 6)])
def test_conv3x3_wgrad_row_pairs_edge_shapes(dev, n, cin, h, w, cout):
    from vistaocr_amd import ops
    x = _rand((n, cin, h, w), 11)
    dy = _rand((n, cout, h, w), 12)
    ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 3, 3), dy.double(), padding=1)
    dw = ops.conv3x3_wgrad(x.to(dev), dy.to(dev))
    _close(dw, ref, 1e-4, 2e-4 * math.sqrt(n * h * w), "conv wgrad (row pairs)")


def test_conv3x3_wgrad_row_pairs_long_stream_matches_direct_kernel(dev):
    """More than 96 segments per workgroup: the slot table's 64-segment ring in LDS is refilled (half a ring at a time) several times.
    16 (ci, co) tiles -> 16 splits of 98 segments each; checked against the direct (non-transform) kernel of conv.hip on the GPU."""
    from vistaocr_amd import ops, _lib
    from vistaocr_amd._lib import call
    lib = _lib.load()
    n, cin, h, w, cout = 32, 256, 14, 220, 256
    g = torch.Generator().manual_seed(5)
    x = (torch.rand((n, cin, h, w), generator=g) * 2 - 1).to(dev)
    dy = (torch.rand((n, cout, h, w), generator=g) * 2 - 1).to(dev)
    dw = ops.conv3x3_wgrad(x, dy)
    ref = torch.empty(cout, cin, 3, 3, dtype=torch.float32, device=dev)
    ws = torch.empty(lib.vocr_conv3x3_wgrad_workspace_bytes(n, cin, h, w, cout) // 4 + 4, dtype=torch.float32, device=dev)
    call("vocr_conv3x3_wgrad", x.data_ptr(), dy.data_ptr(), ref.data_ptr(), ws.data_ptr(), n, cin, h, w, cout, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    _close(dw, ref, 1e-4, 2e-4 * math.sqrt(n * h * w), "conv wgrad (row pairs, long stream)")


def test_conv3x3_pack_is_validated_and_huge_tensors_take_the_direct_rows(dev):
    """Advisor, round 4: (1) conv3x3_forward accepted any 21- or 27-row pack whatever format the library assumes for this channel count (an
    out-of-bounds device read instead of an error); (2) the F(4,3) launches hard-failed for tensors of 2^29 elements or more.  Now the row
    count is checked against the library's own answer, and a tensor beyond the minimal-filtering kernels' 32-bit offsets runs the direct
    kernel on the direct-form rows that trail every transformed pack - checked here against the normal path on one image of the batch."""
    from vistaocr_amd import ops, _lib
    lib = _lib.load()
    cin, cout = 16, 128
    wt = _rand((cout, cin, 3, 3), 2, 0.2).to(dev)
    bias = _rand((cout,), 3).to(dev)
    pf, _ = ops.conv3x3_pack(wt)
    rows = lib.vocr_conv3x3_wino_pack_floats(cout, cin) // (cout * cin)
    assert pf.shape[0] == rows * cin and rows in (21, 27)
    other = 21 if rows == 27 else 27
    xs = _rand((1, cin, 6, 40), 1).to(dev)
    with pytest.raises(ValueError, match="weight pack"):
        ops.conv3x3_forward(xs, torch.zeros(other * cin, cout, device=dev), bias, cout)
    n, h, w = 8, 30, 17500                                   # n * cout * h * w = 5.4e8 >= 2^29
    assert n * cout * h * w >= (1 << 29)
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(n, cin, h, w, generator=g) - 0.5).to(dev)
    y = ops.conv3x3_forward(x, pf, bias, cout)               # direct kernel on the trailing rows
    y3 = ops.conv3x3_forward(x[3:4].contiguous(), pf, bias, cout)      # minimal-filtering kernel
    d = float((y[3:4] - y3).abs().max())
    assert d <= 1e-4 * float(y3.abs().max()), d
    del x, y, y3
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n,h,w,cout", [(2, 30, 70, 64), (3, 30, 601, 64), (2, 60, 257, 16), (1, 5, 3, 7), (4, 7, 1024, 32)])
def test_conv3x3_one_input_channel_kernels(dev, n, h, w, cout):
    """Round 5: the vector-arithmetic kernels of the one-input-channel layers (the first layer of a grey-line model, configs[4]'s rapid_ds
    stage) against F.conv2d / conv2d_weight on the CPU: forward in fp32 and with fp16-rounded operands, weight gradient exact fp32 and
    bitwise reproducible; widths that are not a multiple of 4, rows shorter than a quad, many rows per split."""
    from vistaocr_amd import ops
    x = _rand((n, 1, h, w), 1)
    wt = _rand((cout, 1, 3, 3), 2, 0.3)
    bias = _rand((cout,), 3)
    dy = _rand((n, cout, h, w), 4)
    y = ops.conv3x3_c1_forward(x.to(dev), wt.to(dev), bias.to(dev))
    _close(y, F.conv2d(x, wt, bias, padding=1), 1e-5, 2e-5, "conv Cin=1 fwd")
    yh = ops.conv3x3_c1_forward(x.to(dev), wt.to(dev), bias.to(dev), f16=True)
    _close(yh, F.conv2d(x.half().float(), wt.half().float(), bias, padding=1), 1e-5, 2e-5, "conv Cin=1 fwd, fp16-rounded operands")
    from vistaocr_amd import _lib
    from vistaocr_amd._lib import call
    lib = _lib.load()

    def c1_wgrad():          # the C entry point directly (ops.conv3x3_wgrad picks it for fewer than 32 output channels only)
        dw_ = torch.empty(cout, 1, 3, 3, device=dev)
        ws = torch.empty(lib.vocr_conv3x3_c1_wgrad_workspace_bytes(n, h, cout) // 4 + 4, device=dev)
        call("vocr_conv3x3_c1_wgrad", xd.data_ptr(), dyd.data_ptr(), dw_.data_ptr(), db_.data_ptr(), ws.data_ptr(), n, h, w, cout,
             torch.cuda.current_stream().cuda_stream)
        return dw_
    xd, dyd = x.to(dev), dy.to(dev)
    db_ = torch.empty(cout, device=dev)
    dw = c1_wgrad()
    _close(db_, dy.sum(dim=(0, 2, 3)), 1e-4, 2e-4 * math.sqrt(n * h * w), "conv Cin=1 bias gradient from the same pass")
    _close(dw, torch.nn.grad.conv2d_weight(x, wt.shape, dy, padding=1), 1e-4, 2e-4 * math.sqrt(n * h * w), "conv Cin=1 wgrad")
    assert torch.equal(dw, c1_wgrad())
    _close(ops.conv3x3_wgrad(xd, dyd), dw.cpu(), 1e-4, 2e-4 * math.sqrt(n * h * w), "conv Cin=1 wgrad through ops")
    # the layer op takes these kernels when nothing flows back into the image, the packed kernels when the input needs a gradient
    g = [t.clone().to(dev).requires_grad_(True) for t in (wt, bias)]
    gam, bet = torch.ones(cout, device=dev, requires_grad=True), torch.zeros(cout, device=dev, requires_grad=True)
    rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    outs = []
    for xin in (x.to(dev), x.to(dev).requires_grad_(True)):
        a = ops.ConvBnReluFn.apply(xin, g[0], g[1], gam, bet, rm.clone(), rv.clone(), True, 1e-5, 0.1)
        outs.append(a.detach())
    _close(outs[0], outs[1].cpu(), 1e-4, 1e-4, "Cin=1 layer: vector kernel vs packed kernel")


@pytest.mark.parametrize("n,cin,h,w,cout", [(2, 1, 30, 70, 64), (2, 64, 15, 45, 128), (1, 128, 7, 33, 256), (1, 16, 9, 40, 64), (1, 24, 12, 31, 16),
                                            # the NHWC-fp16 kernel (Cin % 16 == 0) at sizes with several workgroups, partial segments and both channel tilings
                                            (3, 64, 30, 131, 64), (2, 256, 7, 294, 256), (4, 128, 15, 97, 128), (2, 16, 30, 100, 64), (1, 48, 5, 33, 80),
                                            # tile shapes of the forward kernel that only larger launches pick on a 256-CU chip: 16 rows x 32 pixels
                                            # (tiles(8 rows) x 2 channel tiles = 320 > 256 >= 160 = tiles(16 rows) x 2) and 12 rows (384 > 256 >= 256)
                                            (10, 32, 16, 256, 256), (8, 16, 24, 256, 256),
                                            # ... and 8 rows x 64 pixels: a 7-row image whose 320 tiles of 8 x 32 would need two rounds, 160 of 8 x 64 one
                                            (20, 16, 7, 250, 256)])
def test_conv3x3_f16_operands(dev, n, cin, h, w, cout):
    """fp16-operand MFMA conv (config 5): products of fp16-rounded operands, fp32 accumulation — compared with the same
    rounding done on the CPU, so the only difference left is the summation order."""
    from vistaocr_amd import ops
    x = _rand((n, cin, h, w), 1)
    wt = _rand((cout, cin, 3, 3), 2, 0.2)
    bias = _rand((cout,), 3)
    dy = _rand((n, cout, h, w), 4)
    xh, wh, dyh = x.half().float(), wt.half().float(), dy.half().float()
    yr = F.conv2d(xh, wh, bias, padding=1)
    dxr = torch.nn.grad.conv2d_input(x.shape, wh, dyh, padding=1)
    pf, pd = ops.conv3x3_pack_f16(wt.to(dev))
    y = ops.conv3x3_forward_f16(x.to(dev), pf, bias.to(dev), cout)
    _close(y, yr, 1e-5, 2e-5 * (cin * 9) ** 0.5, "conv f16 fwd")
    dx = ops.conv3x3_forward_f16(dy.to(dev), pd, None, cin)
    _close(dx, dxr, 1e-5, 2e-5 * (cout * 9) ** 0.5, "conv f16 dgrad")
    if cin > 3:
        # the fp16-operand weight-gradient kernel (the C entry point) against the same rounding on the CPU ...
        from vistaocr_amd import _lib
        from vistaocr_amd._lib import call
        lib = _lib.load()
        dwr = torch.nn.grad.conv2d_weight(xh, wt.shape, dyh, padding=1)
        dw = torch.empty(cout, cin, 3, 3, device=dev)
        ws = torch.empty(lib.vocr_conv3x3_wgrad_workspace_bytes(n, cin, h, w, cout) // 4 + 4, device=dev)
        xd, dyd = x.to(dev), dy.to(dev)
        call("vocr_conv3x3_wgrad_f16", xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), ws.data_ptr(), n, cin, h, w, cout, torch.cuda.current_stream().cuda_stream)
        _close(dw, dwr, 1e-4, 2e-4 * math.sqrt(n * h * w), "conv f16 wgrad")
        # ... what the fp16 configuration runs without the channel-major copies: the fp32 row-pair kernel on the exact operands ...
        dw32 = ops.conv3x3_wgrad(xd, dyd, f16=True)
        _close(dw32, torch.nn.grad.conv2d_weight(x, wt.shape, dy, padding=1), 1e-4, 2e-4 * math.sqrt(n * h * w), "conv wgrad in the fp16 configuration")
        # ... and with them (round 5): the all-DMA fp16 kernel on [N][C][H][WP] copies with zero-padded rows
        if ops.wgrad_f16_layouts_ok(cin, cout):
            x16p, dy16p = ops.f16_layouts(xd, False, True)[1], ops.f16_layouts(dyd, False, True)[1]
            wp = x16p.shape[3]
            assert wp % 8 == 0 and wp >= w + 8 and float(x16p[..., w:].abs().max()) == 0.0 and torch.equal(x16p[..., :w].float().cpu(), xh)
            dwh = ops.conv3x3_wgrad(xd, dyd, f16=True, x16p=x16p, dy16p=dy16p)
            _close(dwh, dwr, 1e-4, 2e-4 * math.sqrt(n * h * w), "conv f16 wgrad (channel-major copies)")
            dwh2 = ops.conv3x3_wgrad(xd, dyd, f16=True, x16p=x16p, dy16p=dy16p)
            assert torch.equal(dwh, dwh2), "fixed-order slab sum: bitwise reproducible"
    # and it really is fp16 rounding: it differs from the fp32 conv by about 2^-11 relative, not more
    y32 = F.conv2d(x, wt, bias, padding=1)
    rel = float((y.cpu() - y32).abs().max() / y32.abs().max())
    assert rel < 5e-3


@pytest.mark.parametrize("n,c,h,w", [(4, 64, 30, 50), (3, 128, 7, 33)])
def test_conv_bn_relu_fn(dev, n, c, h, w):
    from vistaocr_amd import ops
    cin = 8
    x = _rand((n, cin, h, w), 1)
    wt = _rand((c, cin, 3, 3), 2, 0.3)
    bias = _rand((c,), 3)
    gamma = _rand((c,), 5) * 0.3 + 1.0
    beta = _rand((c,), 6) * 0.2
    rm, rv = _rand((c,), 7) * 0.1, _rand((c,), 8) * 0.2 + 1.0
    da = _rand((n, c, h, w), 9)
    for training in (True, False):
        leaf = [t.clone().requires_grad_(True) for t in (x, wt, bias, gamma, beta)]
        rm_r, rv_r = rm.clone(), rv.clone()
        yr = F.relu(F.batch_norm(F.conv2d(leaf[0], leaf[1], leaf[2], padding=1), rm_r, rv_r, leaf[3], leaf[4],
                                 training=training, momentum=0.1, eps=1e-5))
        gl = [t.clone().to(dev).requires_grad_(True) for t in (x, wt, bias, gamma, beta)]
        rm_g, rv_g = rm.clone().to(dev), rv.clone().to(dev)
        y = ops.ConvBnReluFn.apply(gl[0], gl[1], gl[2], gl[3], gl[4], rm_g, rv_g, training, 1e-5, 0.1)
        _close(y, yr, 1e-4, 2e-4, "conv-bn-relu fwd (training=%s)" % training)
        if training:
            _close(rm_g, rm_r, 1e-5, 1e-6, "running_mean")
            _close(rv_g, rv_r, 1e-5, 1e-6, "running_var")
            yr.backward(da)
            y.backward(da.to(dev))
            names = ["dx", "dw", "dbias", "dgamma", "dbeta"]
            for nm, a, b in zip(names, gl, leaf):
                if nm == "dbias":       # mathematically zero through batch-stat BN; both sides hold rounding noise
                    assert float(a.grad.abs().max()) < 1e-2
                    continue
                # a pre-activation within rounding of 0 may take the other ReLU branch than ATen's: such a flip
                # perturbs the 9*Cin dx entries / one dw row around it, so a tiny mismatch fraction is allowed
                _close(a.grad, b.grad, 2e-3, 2e-4 * float(b.grad.abs().max()) + 1e-5, nm, max_bad_frac=0.02)


@pytest.mark.parametrize("n,c,h,w", [(3, 64, 30, 200), (2, 128, 15, 141), (2, 16, 9, 37)])
def test_conv_bn_relu_pool_fused_backward(dev, n, c, h, w):
    """Conv -> BatchNorm -> ReLU -> FractionalMaxPool2d as ONE layer op (pool fused into the BN pass forward; pooling gradient,
    ReLU and BatchNorm backward fused into one pass backward) against the same chain on PyTorch-CPU with the same pool samples;
    samples include 0 and the largest float32 below 1 (the last two windows then coincide in float32 interval arithmetic)."""
    import os
    from vistaocr_amd import ops
    cin = 8
    x = _rand((n, cin, h, w), 1)
    wt = _rand((c, cin, 3, 3), 2, 0.3)
    bias = _rand((c,), 3)
    gamma = _rand((c,), 5) * 0.3 + 1.0
    beta = _rand((c,), 6) * 0.2
    oh, ow = int(math.floor(h * 0.5)), int(math.floor(w * 0.7))
    g = torch.Generator().manual_seed(11)
    u = torch.rand(n, c, 2, generator=g)
    u[0, 0] = torch.tensor([0.0, 0.0])
    u[0, 1] = torch.tensor([np.float32(1.0) - np.float32(2.0 ** -24), np.float32(1.0) - np.float32(2.0 ** -24)])
    u[1, 0] = torch.tensor([0.999999, 0.5])
    dout = _rand((n, c, oh, ow), 9)
    leaf = [t.clone().requires_grad_(True) for t in (x, wt, bias, gamma, beta)]
    a = F.relu(F.batch_norm(F.conv2d(leaf[0], leaf[1], leaf[2], padding=1), None, None, leaf[3], leaf[4], training=True, eps=1e-5))
    yr = F.fractional_max_pool2d(a, 2, output_size=(oh, ow), _random_samples=u)
    yr.backward(dout)
    results = {}
    for fused in ("1", "0"):
        os.environ["VOCR_POOL_BWD_FUSED"] = fused
        os.environ["VOCR_EXPERIMENTS"] = "1"       # experiment switches are read only then (vistaocr_amd/ops.py:_exp)
        gl = [t.clone().to(dev).requires_grad_(True) for t in (x, wt, bias, gamma, beta)]
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        y = ops.ConvBnReluFn.apply(gl[0], gl[1], gl[2], gl[3], gl[4], rm, rv, True, 1e-5, 0.1, False, u.to(dev), oh, ow)
        y.backward(dout.to(dev))
        results[fused] = (y.detach().cpu(), [t.grad.detach().cpu() for t in gl])
    os.environ.pop("VOCR_POOL_BWD_FUSED", None)
    os.environ.pop("VOCR_EXPERIMENTS", None)
    y1, g1 = results["1"]
    y0, g0 = results["0"]
    _close(y1, yr, 1e-4, 2e-4, "pooled forward")
    for nm, a1, a0, ref in zip(["dx", "dw", "dbias", "dgamma", "dbeta"], g1, g0, [t.grad for t in leaf]):
        if nm == "dbias":
            assert float(a1.abs().max()) < 1e-2 and float(a0.abs().max()) < 1e-2
            continue
        _close(a1, a0, 1e-5, 1e-6 * float(a0.abs().max()) + 1e-7, nm + " fused vs two-pass")
        _close(a1, ref, 2e-3, 2e-4 * float(ref.abs().max()) + 1e-5, nm + " vs torch", max_bad_frac=0.02)


def test_fracpool_bit_exact(dev):
    from vistaocr_amd import ops
    for (n, c, h, w), seed in [((3, 5, 30, 600), 0), ((2, 7, 15, 420), 1), ((2, 4, 15, 47), 2), ((1, 3, 30, 15), 3)]:
        x = _rand((n, c, h, w), seed)
        x[0, 0, 2:4, 5:9] = 0.5
        u = torch.rand(n, c, 2, generator=torch.Generator().manual_seed(seed + 10))
        oh, ow = math.floor(h * 0.5), math.floor(w * 0.7)
        xr = x.clone().requires_grad_(True)
        ref, idx = F.fractional_max_pool2d(xr, 2, output_size=(oh, ow), _random_samples=u, return_indices=True)
        xg = x.to(dev).requires_grad_(True)
        out = ops.FracPoolFn.apply(xg, u.to(dev), oh, ow)
        assert torch.equal(out.cpu(), ref), "fracpool values differ"
        dout = _rand(ref.shape, seed + 20)
        ref.backward(dout)
        out.backward(dout.to(dev))
        assert torch.equal(xg.grad.cpu(), xr.grad), "fracpool backward differs"


def test_relu_maxpool(dev):
    from vistaocr_amd import ops
    x = _rand((2, 1, 60, 122), 0)
    wt = _rand((16, 1, 3, 3), 1, 0.4)
    b = _rand((16,), 2)
    leaf = [t.clone().requires_grad_(True) for t in (x, wt, b)]
    ref = F.max_pool2d(F.relu(F.conv2d(leaf[0], leaf[1], leaf[2], padding=1)), 2, stride=2)
    gl = [t.clone().to(dev).requires_grad_(True) for t in (x, wt, b)]
    out = ops.ConvReluPoolFn.apply(*gl)
    _close(out, ref, 1e-5, 1e-5, "rds fwd")
    do = _rand(ref.shape, 3)
    ref.backward(do)
    out.backward(do.to(dev))
    for nm, a, r in zip(("dx", "dw", "db"), gl, leaf):
        _close(a.grad, r.grad, 1e-4, 1e-4 * float(r.grad.abs().max()) + 1e-6, "rds " + nm)


# the last two shapes have more tiles than one round of resident workgroups, so their last partial round is cut along K into
# tail-filling pieces (gemm.hip, ragged grid): 33x32 = 1056 tiles of 128x64 (288 cut in 2), 1100 tiles of 128x128 (76 cut in 6)
@pytest.mark.parametrize("m,n,k", [(100, 96, 1024), (9408, 128, 1792), (300, 2048, 128), (257, 166, 96), (64, 64, 4000), (1200, 384, 96),
                                   (4200, 2048, 256), (6400, 2816, 384)])
def test_gemm_variants(dev, m, n, k):
    from vistaocr_amd import ops
    a = _rand((m, k), 1)
    b = _rand((k, n), 2)
    bias = _rand((n,), 3)
    ref = a.double() @ b.double()
    tol = 3e-6 * k
    c = torch.empty(m, n, device=dev)
    ops.gemm(0, 0, m, n, k, a.to(dev), k, b.to(dev), n, c, n)
    _close(c, ref, 1e-5, tol, "gemm NN")
    ops.gemm(0, 1, m, n, k, a.to(dev), k, b.t().contiguous().to(dev), k, c, n, bias=bias.to(dev), relu=True)
    _close(c, torch.relu(ref + bias.double()), 1e-5, tol, "gemm NT bias relu")
    ops.gemm(1, 0, m, n, k, a.t().contiguous().to(dev), m, b.to(dev), n, c, n)
    _close(c, ref, 1e-5, tol, "gemm TN")
    ops.gemm(1, 1, m, n, k, a.t().contiguous().to(dev), m, b.t().contiguous().to(dev), k, c, n)
    _close(c, ref, 1e-5, tol, "gemm TT")
    c0 = _rand((m, n), 4)
    c = c0.clone().to(dev)
    ops.gemm(0, 0, m, n, k, a.to(dev), k, b.to(dev), n, c, n, accumulate=True)
    _close(c, ref + c0.double(), 1e-5, tol, "gemm accumulate")
    _close(ops.colsum(a.to(dev)), a.double().sum(0), 1e-5, 1e-5 * m, "colsum")


@pytest.mark.parametrize("T,B,H,din", [(294, 32, 512, 1024), (37, 5, 64, 128)])
def test_gemm_pair_on_the_steps_strided_views_fp64(dev, T, B, H, din):
    """vocr_gemm_pair produces every LSTM data gradient and weight gradient of a step.  The three call shapes of
    vistaocr_amd/ops.py's BiLSTM backward - including the recurrent weight gradient's operands, which are VIEWS (dg[0][B:] is a row
    offset, y[B:, H:] a column offset with ld = 2H) and a K that is not a multiple of the K-tile - against float64 at ~1e-5 of the
    result's scale: a dropped K tail, a missing split-K slab or one wrong leading dimension is a percent-level error."""
    from vistaocr_amd import ops
    G, M = 4 * H, T * B
    dg = _rand((2, M, G), 1, 0.5).to(dev)
    x = _rand((M, din), 2).to(dev)
    y = _rand((M, 2 * H), 3).to(dev)
    wf, wr = _rand((G, din), 4, 0.1).to(dev), _rand((G, din), 5, 0.1).to(dev)
    dgd, xd, yd = dg.double().cpu(), x.double().cpu(), y.double().cpu()

    def close(got, want, what):
        scale = float(want.abs().max())
        err = float((got.double().cpu() - want).abs().max())
        assert err <= 2e-5 * scale, "%s: max abs error %.3e at scale %.3e" % (what, err, scale)

    # data gradient: ONE product whose K runs through both directions' pairs
    dx = torch.empty(M, din, device=dev)
    ops.gemm_pair(1, 0, 0, M, din, G, dg[0], dg[1], G, wf, wr, din, dx, None, din)
    close(dx, dgd[0] @ wf.double().cpu() + dgd[1] @ wr.double().cpu(), "dx = dg_f W_f + dg_r W_r")
    # input weight gradients: two TN products, K = T*B
    dwf, dwr = torch.empty(G, din, device=dev), torch.empty(G, din, device=dev)
    ops.gemm_pair(0, 1, 0, G, din, M, dg[0], dg[1], G, x, x, din, dwf, dwr, din)
    close(dwf, dgd[0].t() @ xd, "dW_ih forward")
    close(dwr, dgd[1].t() @ xd, "dW_ih reverse")
    # recurrent weight gradients on views: forward h_{t-1} = y[t-1, :, :H] against dg[t], reverse h_{t+1} = y[t+1, :, H:] against dg[t]
    m = (T - 1) * B
    dhf, dhr = torch.empty(G, H, device=dev), torch.empty(G, H, device=dev)
    ops.gemm_pair(0, 1, 0, G, H, m, dg[0][B:], dg[1], G, y, y[B:, H:], 2 * H, dhf, dhr, H)
    close(dhf, dgd[0][B:].t() @ yd[:m, :H], "dW_hh forward (row-offset view)")
    close(dhr, dgd[1][:m].t() @ yd[B:, H:], "dW_hh reverse (column-offset view, ld = 2H)")
    # x-projections: two NT products with bias
    bf, br = _rand((G,), 6).to(dev), _rand((G,), 7).to(dev)
    xp = torch.empty(2, M, G, device=dev)
    ops.gemm_pair(0, 0, 1, M, G, din, x, x, din, wf, wr, din, xp[0], xp[1], G, bias0=bf, bias1=br)
    close(xp[0], xd @ wf.double().cpu().t() + bf.double().cpu(), "x-projection forward")
    close(xp[1], xd @ wr.double().cpu().t() + br.double().cpu(), "x-projection reverse")


def test_permute_and_elementwise(dev):
    from vistaocr_amd import ops
    x = _rand((3, 16, 7, 37), 0)
    xg = x.to(dev).requires_grad_(True)
    out = ops.PermuteBchwToWbchFn.apply(xg)
    ref = x.permute(3, 0, 1, 2).contiguous().view(-1, 16 * 7)
    assert torch.equal(out.cpu(), ref)
    g = _rand(ref.shape, 1)
    out.backward(g.to(dev))
    assert torch.equal(xg.grad.cpu(), g.view(37, 3, 16, 7).permute(1, 2, 3, 0).contiguous())
    a = _rand((1000,), 2).to(dev)
    o = ops.DropoutFn.apply(a, 0.5, 123)
    frac = float((o == 0).float().mean())
    assert 0.4 < frac < 0.6
    kept = o != 0
    assert torch.allclose(o[kept], a[kept] * 2)


def _ref_bilstm(x, lens, params, H):
    T, B, D = x.shape
    m = torch.nn.LSTM(D, H, num_layers=1, bidirectional=True)
    with torch.no_grad():
        for name, p in zip(["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse",
                            "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse"], params):
            getattr(m, name).copy_(p)
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, lens)
    out, _ = m(packed)
    y, _ = torch.nn.utils.rnn.pad_packed_sequence(out)
    return m, y


@pytest.mark.parametrize("T,B,D,H,lens", [(20, 4, 32, 48, [20, 17, 9, 1]), (37, 32, 128, 256, None), (12, 40, 64, 32, None),
                                          (9, 1, 16, 16, [9]), (26, 22, 64, 512, None), (14, 6, 32, 512, None)])
def test_bilstm_layer(dev, T, B, D, H, lens):
    from vistaocr_amd import ops
    if lens is None:
        lens = sorted([max(1, T - (i * T) // (B + 3)) for i in range(B)], reverse=True)
        lens[0] = T
    x = _rand((T, B, D), 1)
    for b in range(B):
        x[lens[b]:, b] = 0.37                         # junk in the padding must not leak
    params = [_rand(s, 10 + i, 0.25) for i, s in enumerate([(4 * H, D), (4 * H, H), (4 * H,), (4 * H,)] * 2)]
    xr = x.clone().requires_grad_(True)
    m, yr = _ref_bilstm(xr, lens, params, H)
    dy = _rand(yr.shape, 30)
    yr.backward(dy)
    xg = x.reshape(T * B, D).to(dev).requires_grad_(True)
    pg = [p.clone().to(dev).requires_grad_(True) for p in params]
    lens_dev = torch.tensor(lens, dtype=torch.int32, device=dev)
    y = ops.BiLstmLayerFn.apply(xg, lens_dev, T, B, *pg)
    _close(y.view(T, B, 2 * H), yr, 1e-4, 2e-5, "lstm fwd")
    y.backward(dy.reshape(T * B, 2 * H).to(dev))
    gx_ref = xr.grad.clone()
    for b in range(B):
        gx_ref[lens[b]:, b] = 0
    _close(xg.grad.view(T, B, D), gx_ref, 1e-3, 2e-5 * float(gx_ref.abs().max()) + 1e-6, "lstm dx")
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse",
             "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
    for nm, p in zip(names, pg):
        r = getattr(m, nm).grad
        _close(p.grad, r, 1e-3, 5e-5 * float(r.abs().max()) + 1e-6, "lstm d" + nm)


@pytest.mark.parametrize("T,B,H", [(40, 20, 128), (25, 5, 64), (33, 32, 512), (12, 40, 256), (9, 64, 128), (21, 27, 256), (15, 7, 512),
                                   (1, 1, 256), (2, 3, 512), (3, 64, 512), (19, 21, 512), (7, 17, 512)])
def test_persistent_sweeps_match_per_step_launches(dev, T, B, H):
    """Every kind of persistent sweep (VOCR_LSTM_SWEEP: self-validating hand-off on 4-row chains - wide members at 16 < B <= 32,
    H = 512 -, arrival flags on 16-row chains; each also with the write-through hand-off of a chain spread over XCDs) against
    one launch per step: forward (y, gates, cell) and backward (dgates), bit-identical where the summation order is the same, to
    rounding where it is another fixed order."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import torch, sys; sys.path.insert(0, %r)\n"
        "from vistaocr_amd import _lib, ops\n"
        "from vistaocr_amd._lib import call\n"
        "lib = _lib.load(); dev = torch.device('cuda:0'); T, B, H = %d, %d, %d\n"
        "g = torch.Generator().manual_seed(0)\n"
        "xp = (torch.rand(2, T * B, 4 * H, generator=g) - 0.5).to(dev); wf = ((torch.rand(4 * H, H, generator=g) - 0.5) * 0.3).to(dev)\n"
        "wr = ((torch.rand(4 * H, H, generator=g) - 0.5) * 0.3).to(dev); dy = (torch.rand(T * B, 2 * H, generator=g) - 0.5).to(dev)\n"
        "lens = torch.tensor(sorted([max(1, T - 2 * i) for i in range(B)], reverse=True), dtype=torch.int32, device=dev)\n"
        "y = torch.empty(T * B, 2 * H, device=dev); gt = torch.empty(2, T * B, 4 * H, device=dev); c = torch.empty(2, T * B, H, device=dev)\n"
        "dg = torch.full((2, T * B, 4 * H), float('nan'), device=dev)\n"
        "ws = torch.zeros(lib.vocr_lstm_workspace_bytes(T, B, H) // 4 + 16, device=dev); s = torch.cuda.current_stream().cuda_stream\n"
        "hw = torch.zeros(4, dtype=torch.int32, device=dev)\n"          # the caller's health word: a hand-off time-out is reported there
        "call('vocr_lstm_fwd', xp.data_ptr(), wf.data_ptr(), wr.data_ptr(), lens.data_ptr(), y.data_ptr(), gt.data_ptr(), c.data_ptr(), ws.data_ptr(), T, B, H, hw.data_ptr(), s)\n"
        "if H >= 128:\n"
        "    wtf, wtr = ops.transpose2d(wf), ops.transpose2d(wr)\n"
        "    call('vocr_lstm_bwd', dy.data_ptr(), wtf.data_ptr(), wtr.data_ptr(), lens.data_ptr(), gt.data_ptr(), c.data_ptr(), dg.data_ptr(), ws.data_ptr(), T, B, H, hw.data_ptr(), s)\n"
        "torch.cuda.synchronize(); st = int(hw[0])\n"
        "print('STATUS', st); torch.save((y.cpu(), gt.cpu(), c.cpu(), dg.cpu()), sys.argv[1])\n"
    ) % (root, T, B, H)
    outs = []
    # per-step | default (newest kind the shape fits) | 4-row chains without wide members | 16-row flag chains | the same with the
    # write-through hand-off
    modes = ("step", "wide4", "chain4", "chain16", "wide4/wt", "chain4/wt", "chain16/wt")
    for mode_ in modes:
        mode = mode_.split("/")[0]
        f = tempfile.mktemp(suffix=".pt")
        r = subprocess.run([sys.executable, "-c", code, f, mode], env=dict(os.environ, VOCR_LSTM_SWEEP=mode, VOCR_LSTM_WRITE_THROUGH="1" if "/wt" in mode_ else "0"),
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-500:]
        assert "STATUS 0" in r.stdout, r.stdout
        outs.append(torch.load(f))
        os.unlink(f)
    eight_row = H in (256, 512) and B <= 32          # mode 3 then runs the 4x4x1 kernels: another (fixed) summation order
    for mode_, other in zip(modes[1:], outs[1:]):
        mode = mode_.split("/")[0]
        for nm, a, b in zip(("y", "gates", "cell", "dgates"), outs[0], other):
            if H < 128 and nm == "dgates":
                continue                      # no backward fast path below H = 128: dgates untouched in both runs
            reordered = (mode != "chain16" and eight_row) or nm == "dgates"       # 4x4x1 kernels / K-owner backward
            if reordered:
                # same arithmetic, different fp32 summation order: rounding differences only
                tol = 3e-5 * float(a.abs().max())
                assert float((a - b).abs().max()) <= tol, "%s (mode %s): max |diff| %.3e > %.3e" % (nm, mode_, float((a - b).abs().max()), tol)
                continue
            assert torch.equal(a, b), "%s (mode %s) differs: max |diff| %.3e at %d of %d" % (nm, mode_, float((a - b).abs().max()), int((a != b).sum()), a.numel())


@pytest.mark.parametrize("n,c,h,w,pooled", [(4, 64, 30, 50, False), (3, 128, 7, 33, False), (32, 8, 30, 600, False), (2, 16, 15, 42, True), (3, 8, 30, 76, True)])
def test_bn_two_launch_form_is_bit_identical_to_three(dev, n, c, h, w, pooled, monkeypatch):
    """Round 4 folds the "final" step of the BatchNorm statistics (forward) and of its two backward sums into the pass that consumes
    them (every workgroup adds its channel's chunks in chunk order).  Same arithmetic, same order: outputs, saved statistics, running
    statistics, num_batches_tracked and all gradients must equal the three-launch form bit for bit - also with more than one chunk per
    channel (32 x 30 x 600: 36 chunks)."""
    from vistaocr_amd import ops
    g = torch.Generator().manual_seed(11)
    cin = 4
    x = (torch.rand(n, cin, h, w, generator=g) - 0.4).to(dev)
    wt = ((torch.rand(c, cin, 3, 3, generator=g) - 0.5) * 0.3).to(dev)
    bias = (torch.rand(c, generator=g) - 0.5).to(dev)
    gamma = (torch.rand(c, generator=g) * 0.5 + 0.75).to(dev)
    beta = ((torch.rand(c, generator=g) - 0.5) * 0.4).to(dev)
    oh, ow = (int(h * 0.5), int(w * 0.7)) if pooled else (0, 0)
    samples = torch.rand(n, c, 2, generator=g).to(dev) if pooled else None
    dout = torch.rand((n, c, oh, ow) if pooled else (n, c, h, w), generator=g).to(dev) - 0.5
    results = []
    for fused in (False, True):
        monkeypatch.setattr(ops, "_BN_FUSED_FINAL", fused)
        xs = x.clone().requires_grad_(True)
        ws = wt.clone().requires_grad_(True)
        bs, gs, bes = bias.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        out = ops.ConvBnReluFn.apply(xs, ws, bs, gs, bes, rm, rv, True, 1e-5, 0.1, False, samples, oh, ow, nbt, None)
        out.backward(dout)
        results.append([t.detach().cpu() for t in (out, rm, rv, nbt, xs.grad, ws.grad, bs.grad, gs.grad, bes.grad)])
    for name, a, b in zip(("out", "running_mean", "running_var", "num_batches_tracked", "dx", "dw", "dbias", "dgamma", "dbeta"), *results):
        assert torch.equal(a, b), "%s differs between the three-launch and the two-launch form: max |diff| %.3e" % (name, float((a.double() - b.double()).abs().max()))
    assert int(results[1][3]) == 1


# widths of every residue mod 4: the one-pass backward works on whole 16-byte pixel groups (W % 4 == 0: 44, 294); the other widths (a ragged
# batch is as wide as its widest line: configs[3]'s 1178) take the pooling gradient and the BatchNorm backward as separate passes - a
# pixel-by-pixel form of the one-pass kernel measured 1219 against 1243 us for the whole layer backward at 30x1178 and nothing in the step
@pytest.mark.parametrize("n,cin,cout,h,w", [(3, 5, 8, 30, 75), (2, 64, 64, 15, 42), (2, 8, 8, 14, 44), (2, 4, 8, 9, 33), (1, 8, 16, 30, 294)])
def test_fused_bn_relu_fracpool_equals_the_two_separate_passes(dev, n, cin, cout, h, w):
    """ConvBnReluFn with pool_samples (one pass: BN-apply + ReLU + FractionalMaxPool) == ConvBnReluFn then FracPoolFn: bit for bit
    forward (and running statistics); the backward of the fused op takes its two BatchNorm sums from the pooled tensors (another
    summation order, in double), so gradients agree to fp32 rounding."""
    import math
    from vistaocr_amd import ops
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(n, cin, h, w, generator=g) - 0.3).to(dev)
    wt = ((torch.rand(cout, cin, 3, 3, generator=g) - 0.5) * 0.2).to(dev)
    bias = (torch.rand(cout, generator=g) - 0.5).to(dev)
    gamma = (torch.rand(cout, generator=g) - 0.3).to(dev)          # some negative scales: max does not commute with BN
    beta = (torch.rand(cout, generator=g) - 0.5).to(dev)
    u = torch.rand(n, cout, 2, generator=g).to(dev)
    oh, ow = math.floor(h * 0.5), math.floor(w * 0.7)
    dout = (torch.rand(n, cout, oh, ow, generator=g) - 0.5).to(dev)
    res = []
    for fused in (False, True):
        leaves = [t.clone().requires_grad_(True) for t in (x, wt, bias, gamma, beta)]
        rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
        if fused:
            out = ops.ConvBnReluFn.apply(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], rm, rv, True, 1e-5, 0.1, False, u, oh, ow)
        else:
            a = ops.ConvBnReluFn.apply(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], rm, rv, True, 1e-5, 0.1, False)
            out = ops.FracPoolFn.apply(a, u, oh, ow)
        out.backward(dout)
        res.append([out.detach().cpu()] + [t.grad.cpu() for t in leaves] + [rm.cpu(), rv.cpu()])
    for nm, a, b in zip(("out", "dx", "dw", "dbias", "dgamma", "dbeta", "running_mean", "running_var"), res[0], res[1]):
        if nm == "dbias":       # exactly zero in exact arithmetic (bias in front of batch-stat BN): float-atomic rounding noise only
            assert float((a - b).abs().max()) < 1e-3
            continue
        if nm in ("out", "running_mean", "running_var"):
            assert torch.equal(a, b), "%s differs: max |diff| %.3e" % (nm, float((a - b).abs().max()))
        else:
            _close(b, a, 1e-5, 1e-6 * float(a.abs().max()) + 1e-7, nm)


@pytest.mark.parametrize("T,B,H", [(33, 32, 512), (21, 27, 256), (12, 40, 256), (17, 9, 128)])
def test_lstm_backward_bias_gradient_is_the_column_sum_of_dgates(dev, T, B, H):
    """vocr_lstm_bwd_bias: dbias[dir] accumulated inside the sweep (8-row K-owner kernel) or by column sums (other paths)."""
    from vistaocr_amd import _lib, ops
    from vistaocr_amd._lib import call
    lib = _lib.load()
    g = torch.Generator().manual_seed(2)
    xp = (torch.rand(2, T * B, 4 * H, generator=g) - 0.5).to(dev)
    wf = ((torch.rand(4 * H, H, generator=g) - 0.5) * 0.3).to(dev)
    wr = ((torch.rand(4 * H, H, generator=g) - 0.5) * 0.3).to(dev)
    dy = (torch.rand(T * B, 2 * H, generator=g) - 0.5).to(dev)
    lens = torch.tensor(sorted([max(1, T - 2 * i) for i in range(B)], reverse=True), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    y = torch.empty(T * B, 2 * H, device=dev)
    gt = torch.empty(2, T * B, 4 * H, device=dev)
    c = torch.empty(2, T * B, H, device=dev)
    dg = torch.empty(2, T * B, 4 * H, device=dev)
    db = torch.full((2, 4 * H), float("nan"), device=dev)
    ws = torch.zeros(lib.vocr_lstm_workspace_bytes(T, B, H) // 4 + 16, device=dev)
    call("vocr_lstm_fwd", xp.data_ptr(), wf.data_ptr(), wr.data_ptr(), lens.data_ptr(), y.data_ptr(), gt.data_ptr(), c.data_ptr(),
         ws.data_ptr(), T, B, H, None, s)
    wtf, wtr = ops.transpose2d(wf), ops.transpose2d(wr)
    call("vocr_lstm_bwd_bias", dy.data_ptr(), wtf.data_ptr(), wtr.data_ptr(), lens.data_ptr(), gt.data_ptr(), c.data_ptr(),
         dg.data_ptr(), db.data_ptr(), ws.data_ptr(), T, B, H, None, s)
    torch.cuda.synchronize()
    ref = dg.double().sum(dim=1)
    assert float((db.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-6       # fp32 summation-order tolerance


@pytest.mark.parametrize("T,B,H,cut", [(40, 20, 128, 20), (33, 32, 512, 7), (12, 40, 256, 11), (25, 5, 64, 12), (21, 27, 256, 9), (19, 21, 512, 1),
                                       (23, 13, 512, 21)])
def test_lstm_forward_step_ranges_resume_bit_exactly(dev, T, B, H, cut):
    """vocr_lstm_fwd_range: steps [0, cut) then [cut, T) leave exactly what one whole sweep leaves (y, gates, cell)."""
    from vistaocr_amd import _lib
    from vistaocr_amd._lib import call
    lib = _lib.load()
    g = torch.Generator().manual_seed(1)
    xp = (torch.rand(2, T * B, 4 * H, generator=g) - 0.5).to(dev)
    wf = ((torch.rand(4 * H, H, generator=g) - 0.5) * 0.3).to(dev)
    wr = ((torch.rand(4 * H, H, generator=g) - 0.5) * 0.3).to(dev)
    lens = torch.tensor(sorted([max(1, T - 2 * i) for i in range(B)], reverse=True), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    outs = []
    for ranges in (((0, T),), ((0, cut), (cut, T))):
        y = torch.full((T * B, 2 * H), float("nan"), device=dev)
        gt = torch.full((2, T * B, 4 * H), float("nan"), device=dev)
        c = torch.full((2, T * B, H), float("nan"), device=dev)
        ws = torch.zeros(lib.vocr_lstm_workspace_bytes(T, B, H) // 4 + 16, device=dev)
        for a, b in ranges:
            call("vocr_lstm_fwd_range", xp.data_ptr(), wf.data_ptr(), wr.data_ptr(), lens.data_ptr(), y.data_ptr(), gt.data_ptr(),
                 c.data_ptr(), ws.data_ptr(), T, B, H, a, b, None, s)
        torch.cuda.synchronize()
        outs.append((y.cpu(), gt.cpu(), c.cpu()))
    for nm, a, b in zip(("y", "gates", "cell"), outs[0], outs[1]):
        assert not torch.isnan(a).any()
        assert torch.equal(a, b), "%s differs after resuming: max |diff| %.3e" % (nm, float((a - b).abs().max()))


@pytest.mark.parametrize("T,B,V,L", [(30, 4, 20, [5, 3, 1, 0]), (147, 8, 96, None), (60, 3, 166, [29, 10, 2]),
                                     # 2L+1 in (64, 128]: two extended-label positions per lane (ctc_alpha_beta128_kernel) - the seam at position 64
                                     # crossed by odd and even label counts, a short line and an empty one beside the long ones; L = 63 fills all 127
                                     (150, 6, 166, [39, 32, 31, 33, 7, 0]), (140, 3, 96, [63, 62, 40]),
                                     # 2L+1 > 128: the generic LDS kernel
                                     (150, 2, 96, [70, 64])])
def test_ctc_loss_and_grad(dev, T, B, V, L):
    from vistaocr_amd import CTCLoss
    g = torch.Generator().manual_seed(0)
    logits = _rand((T, B, V), 1, 3.0)
    if L is None:
        L = [12] * B
    act = [T - 3 * i for i in range(B)]
    tl = torch.tensor(L, dtype=torch.int32)
    tg = torch.randint(1, V, (int(tl.sum()),), generator=g).to(torch.int32)
    if tg.numel() >= 4:
        tg[1] = tg[0]                                                # repeated label needs the blank in between
        tg[3] = tg[2]
    lr = logits.clone().requires_grad_(True)
    ref = F.ctc_loss(F.log_softmax(lr, 2), tg.long(), torch.tensor(act), tl.long(), blank=0, reduction="sum")
    ref.backward()
    lg = logits.clone().to(dev).requires_grad_(True)
    loss = CTCLoss()(lg, tg, torch.tensor(act, dtype=torch.int32), tl)
    assert tuple(loss.shape) == (1,)
    lv, rv = float(loss.detach()), float(ref.detach())
    assert abs(lv - rv) <= 1e-5 * abs(rv) + 1e-4, (lv, rv)
    loss.backward()
    _close(lg.grad, lr.grad, 1e-3, 2e-5, "ctc dlogits")


def test_argmax_and_collapse(dev):
    from vistaocr_amd import ops, english_alphabet
    from vistaocr_amd.decoder import greedy_labels_device, greedy_label_sequences
    x = _rand((50, 7, 96), 0)
    x[3, 2, 10] = x[3, 2, 50] = 5.0                                 # tie: first index wins
    x[4, 1, :] = 0.25
    idx, mx = ops.argmax_rows(x.to(dev))
    assert torch.equal(idx.cpu().long(), x.argmax(2))
    assert torch.equal(mx.cpu(), x.max(2)[0])
    al = english_alphabet()
    lens = [50, 50, 44, 30, 12, 2, 1]
    a = greedy_labels_device(x.to(dev), lens, al)
    _, b = greedy_label_sequences(x.to(dev), lens, al)
    assert a == b


def test_clamp_adam_matches_torch(dev):
    from vistaocr_amd import ops
    n = 10007
    p0 = _rand((n,), 0)
    for wd in (0.0, 0.01):
        pr = p0.clone().requires_grad_(True)
        opt = torch.optim.Adam([pr], lr=1e-3, weight_decay=wd)
        pg = p0.clone().to(dev)
        m = torch.zeros(n, device=dev)
        v = torch.zeros(n, device=dev)
        for step in range(1, 6):
            g = _rand((n,), step, 8.0)
            g[:100] *= 1e-7
            pr.grad = g.clone().clamp_(-5, 5)
            opt.step()
            ops.clamp_adam(pg, g.to(dev), m, v, 1e-3, 0.9, 0.999, 1e-8, wd, 5.0, 1.0, step)
        _close(pg, pr, 1e-6, 1e-7, "adam wd=%g" % wd)


@pytest.mark.parametrize("n,p", [(1 << 20, 0.5), (4099, 0.3), (3, 0.5), (1 << 18, 0.0)])
def test_device_drawn_dropout_is_a_function_of_seed_and_element(dev, n, p):
    """nn.LSTM(dropout=p)'s inter-layer mask drawn on the device (vocr_dropout_fwd; no reference stream to match: torch's own Philox
    offsets are not part of the contract).  What IS: mask values are exactly 0 or 1 / (1 - p), out == x * mask bit for bit, the mask is a
    function of (seed, element index) alone - same seed: same mask, whatever the length or the 16-byte / scalar-tail split; another seed:
    another mask - the keep rate sits inside a 5-sigma binomial band, and the backward multiplies by the SAME mask."""
    from vistaocr_amd import ops
    x = (_rand((n,), 31) + 2.0).to(dev)                        # no zeros in x: out == 0 <=> mask == 0

    def draw(t, seed):
        tt = t.clone().requires_grad_(True)
        out = ops.DropoutFn.apply(tt, p, seed)
        out.backward(torch.ones_like(out))
        return out.detach(), tt.grad.detach()

    out, mask = draw(x, 1234)
    scale = 1.0 / (1.0 - p)
    vals = torch.unique(mask).cpu().tolist()
    assert set(vals) <= {0.0, float(np.float32(scale))}, vals
    assert torch.equal(out, x * mask)
    keep = float((mask != 0).double().mean())
    assert abs(keep - (1.0 - p)) <= 5.0 * math.sqrt(max(p * (1.0 - p), 1e-12) / n) + (0.5 if n < 16 else 0.0), keep
    out2, mask2 = draw(x, 1234)
    assert torch.equal(mask, mask2) and torch.equal(out, out2)
    if n > 8:
        # element e's draw does not depend on the tensor's length (so neither on which loop of the kernel handled it)
        _, mask_short = draw(x[: n - 3].contiguous(), 1234)
        assert torch.equal(mask_short, mask[: n - 3])
        if p > 0:
            _, mask3 = draw(x, 1235)
            assert not torch.equal(mask, mask3)
            agree = float((mask3 == mask).double().mean())
            assert abs(agree - (p * p + (1 - p) * (1 - p))) < 0.05, agree      # independent streams
