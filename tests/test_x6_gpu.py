"""GPU: vocr_gemm_x6_split / vocr_gemm_x6 (include/vocr.h) - fp32 products on the bf16 matrix pipe from operands split EXACTLY into three bf16
planes (src/models/cnnlstm.py:143-154: the nn.LSTM projections the reference leaves to cuBLAS).
  * the three planes sum to the fp32 operand exactly (the split is lossless);
  * products against fp64 at an fp32 bar (the f32-MFMA GEMM of the same product is measured beside it: the bf16x6 error must not exceed 2 x its own) on
    ragged shapes (M, N, K not multiples of the tiles), K-contiguous and K-strided sources, two-piece sources along either axis, the dropout mask on
    the read, bias + ReLU, both output cuts, views (row / k16 offsets), the leftover round's K cut, the narrow and the wide tile;
  * run-to-run bit-identical (the slabs are added in a fixed order);
  * the opt-in fp16x3 split (vocr_gemm_h3*): how well its two planes represent the operand, and every product test above at the same bars;
  * the BiLSTM layer op with the split products and with the f32 kernels (VOCR_LSTM_GEMM) against each other at the layer tests' tolerances: dense and packed rows, a batch
    size that is not a multiple of 16 (the recurrent weight gradient then stays on the f32 kernels)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rnd(g, dev, *s, a=1.0):
    return ((torch.rand(*s, generator=g) - 0.5) * a).to(dev)


def _planes(x, rows, k, kc, ld, x2=None, seg=0, axis=0, mask=None, scheme="bf16x6", bound=0.0):
    from vistaocr_amd import ops
    return ops.x6_planes(x, rows, k, kc, ld, x2=x2, seg=seg, axis=axis, mask=mask, scheme=scheme, bound=bound)


SCHEMES = ["bf16x6", "fp16x3"]           # the default split and the opt-in one (include/vocr.h: vocr_gemm_h3*): same product tests, same bars


def test_the_split_is_exact(dev):
    """plane0 + plane1 + plane2 == x bit for bit (fragment order undone on the host), including tiny and huge magnitudes."""
    from vistaocr_amd import _lib
    g = torch.Generator().manual_seed(1)
    rows, k = 70, 48
    x = _rnd(g, dev, rows, k, a=2.0) * torch.pow(10.0, _rnd(g, dev, rows, k, a=30.0))
    buf = _planes(x, rows, k, True, k)
    RT, KK = 8, 4
    p = buf.view(torch.bfloat16).view(3, RT, KK, 64, 8).float().cpu()                    # [plane][row tile][k16 step][lane][8]
    rec = torch.zeros(RT * 32, KK * 16, dtype=torch.float64)
    for lane in range(64):
        r, h = lane & 31, lane >> 5
        blk = p[:, :, :, lane, :].double().sum(0)                    # [RT][KK][8]
        for rt in range(RT):
            for kk in range(KK):
                rec[32 * rt + r, 16 * kk + 8 * h: 16 * kk + 8 * h + 8] = blk[rt, kk]
    assert torch.equal(rec[:rows, :k].float(), x.cpu()), "the three planes do not sum to the operand"
    assert float(rec[rows:].abs().max()) == 0.0 and float(rec[:, k:].abs().max()) == 0.0                  # padding is zeros


@pytest.mark.parametrize("kc", [True, False])
def test_the_fp16x3_split_represents_its_operand(dev, kc):
    """(plane0 + plane1) / scale against x: 2^-22 relative for elements within 2^17 of their row's maximum, 2^-39 of that maximum below; the stored
    maxima are the rows' maxima; the scale brings them into [2^14, 2^15); K-contiguous and K-strided sources (the latter: maxima by a pass of its own)."""
    g = torch.Generator().manual_seed(2)
    rows, k = 70, 200
    x = _rnd(g, dev, rows, k, a=2.0) * torch.pow(10.0, _rnd(g, dev, rows, k, a=8.0)) * torch.pow(10.0, _rnd(g, dev, rows, 1, a=20.0))
    x[5] = 0.0                                                                            # an all-zero row
    src = x if kc else x.t().contiguous()
    buf = _planes(src, rows, k, kc, src.stride(0), scheme="fp16x3")
    RT, KK = 8, 14
    nb = 2 * RT * KK * 64 * 8
    p = buf[:nb].view(torch.float16).view(2, RT, KK, 64, 8).double().cpu()
    amax = buf[nb:].view(torch.float32).cpu()
    assert amax.numel() == RT * 32 and torch.equal(amax[:rows], x.abs().max(1).values.cpu()) and float(amax[rows:].abs().max()) == 0.0
    rec = torch.zeros(RT * 32, KK * 16, dtype=torch.float64)
    for lane in range(64):
        r, h = lane & 31, lane >> 5
        blk = p[:, :, :, lane, :].sum(0)
        for rt in range(RT):
            for kk in range(KK):
                rec[32 * rt + r, 16 * kk + 8 * h: 16 * kk + 8 * h + 8] = blk[rt, kk]
    xd = x.double().cpu()
    am = amax[:rows].double().unsqueeze(1)
    e = torch.floor(torch.log2(torch.clamp(am, min=1e-300)))
    scale = torch.where(am > 0, torch.pow(2.0, 14.0 - e), torch.ones_like(am))
    assert float((am * scale)[am > 0].min()) >= 2.0 ** 14 and float((am * scale).max()) < 2.0 ** 15
    err = (rec[:rows, :k] / scale - xd).abs()
    bound = torch.maximum(xd.abs() * 2.0 ** -22, am * 2.0 ** -39)
    assert bool((err <= bound).all()), float((err / torch.clamp(bound, min=1e-300)).max())
    assert float(rec[rows:].abs().max()) == 0.0 and float(rec[:, k:].abs().max()) == 0.0 and float(rec[5].abs().max()) == 0.0


@pytest.mark.parametrize("kc", [True, False])
def test_fp16x3_split_with_a_callers_bound(dev, kc):
    """bound > 0 stands in for the rows' maxima (no pass over the source): the stored maxima are the bound, the product is the measured-maxima
    product to fp32 accuracy (operands inside the bound: an LSTM output times its dropout scale)."""
    from vistaocr_amd import ops
    g = torch.Generator().manual_seed(11)
    m, n, k = 520, 256, 1000
    A = _rnd(g, dev, m, k, a=2.0) * (torch.rand(m, k, generator=g) > 0.5).float().to(dev) * 2.0          # |a| <= 2: y in (-1, 1) times a 0 / 2 mask
    Bm = _rnd(g, dev, n, k, a=0.3)
    src = A if kc else A.t().contiguous()
    pa = _planes(src, m, k, kc, src.stride(0), scheme="fp16x3", bound=2.0)
    nb = 2 * (3 * 8) * (32 * 2) * 64 * 8
    amax = pa[nb:].view(torch.float32)
    assert torch.equal(amax[:m], torch.full((m,), 2.0, device=dev)) and float(amax[m:].abs().max()) == 0.0
    pb = _planes(Bm, n, k, True, k, scheme="fp16x3")
    k16 = (k + 15) // 16 * 16
    c = torch.empty(m, n, device=dev)
    ops.gemm_x6(pa, m, k, pb, n, k, m, n, k16, c, n)
    ex = A.double() @ Bm.double().T
    ref = torch.empty(m, n, device=dev)
    ops.gemm(0, 1, m, n, k, A, k, Bm, k, ref, n)
    e3, e32 = float((c.double() - ex).abs().max()), float((ref.double() - ex).abs().max())
    assert e3 <= max(2.0 * e32, 2e-7 * float(ex.abs().max())), (e3, e32)


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("m,n,k,kc_a,kc_b", [(300, 200, 80, True, True), (1000, 384, 1024, True, False), (2048, 1024, 2000, False, False),
                                             (9408, 1024, 512, True, True), (520, 4096, 256, True, True), (4096, 512, 4704, False, False),
                                             (1024, 128, 4704, False, False)])          # the last one: four tiles, K cut sixteen ways
def test_products_against_fp64(dev, m, n, k, kc_a, kc_b, scheme):
    from vistaocr_amd import ops
    g = torch.Generator().manual_seed(m + n + k)
    A = _rnd(g, dev, m, k, a=2.0)
    Bm = _rnd(g, dev, n, k, a=0.5)
    bias = _rnd(g, dev, n, a=0.2)
    a_src = A if kc_a else A.t().contiguous()                       # K-strided source: [k][m]
    b_src = Bm if kc_b else Bm.t().contiguous()
    pa = _planes(a_src, m, k, kc_a, a_src.stride(0), scheme=scheme)
    pb = _planes(b_src, n, k, kc_b, b_src.stride(0), scheme=scheme)
    k16 = (k + 15) // 16 * 16
    ex = A.double() @ Bm.double().T + bias.double()
    outs = []
    for _ in range(2):
        c = torch.full((m, n), float("nan"), device=dev)
        ops.gemm_x6(pa, m, k, pb, n, k, m, n, k16, c, n, bias0=bias)
        outs.append(c)
    assert torch.equal(outs[0], outs[1]), "not bit-identical run to run"
    ref = torch.empty(m, n, device=dev)
    ops.gemm(0, 1, m, n, k, A, k, Bm, k, ref, n, bias=bias)
    e6 = float((outs[0].double() - ex).abs().max())
    e32 = float((ref.double() - ex).abs().max())
    scale = float(ex.abs().max())
    assert not torch.isnan(outs[0]).any()
    assert e6 <= max(2.0 * e32, 2e-7 * scale), (e6, e32, scale)


@pytest.mark.parametrize("scheme", SCHEMES)
def test_two_piece_sources_mask_cuts_and_views(dev, scheme):
    from vistaocr_amd import ops
    import functools
    _planes = functools.partial(globals()["_planes"], scheme=scheme)
    g = torch.Generator().manual_seed(9)
    R, G4, D, Hh, sh = 1024, 256, 512, 128, 32
    dg = _rnd(g, dev, 2, R, G4, a=0.1)
    wf, wr = _rnd(g, dev, G4, D, a=0.3), _rnd(g, dev, G4, D, a=0.3)
    x = _rnd(g, dev, R, D, a=2.0)
    mask = (torch.rand(R, D, generator=g) > 0.5).float().to(dev) * 2.0
    y = _rnd(g, dev, R, 2 * Hh, a=1.5)
    b = _rnd(g, dev, 2, G4, a=0.1)
    w = torch.cat([wf, wr], 0)
    # x-projection with the mask on the read, the two weight matrices along the rows, outputs cut along the columns, bias + ReLU
    xa = _planes(x, R, D, True, D, mask=mask)
    wk = _planes(wf, 2 * G4, D, True, D, x2=wr, seg=G4, axis=1)
    out = torch.empty(2, R, G4, device=dev)
    ops.gemm_x6(xa, R, D, wk, 2 * G4, D, R, 2 * G4, D, out[0], G4, c1=out[1], csplit=G4, bias0=b[0], bias1=b[1], relu=True)
    ex = torch.relu((x * mask).double() @ w.double().T + b.reshape(-1).double())
    got = torch.cat([out[0], out[1]], 1).double()
    assert float((got - ex).abs().max()) <= 3e-6 * float(ex.abs().max())
    # data gradient: the two gate planes along K, the weights' transpose from the two matrices along K
    da = _planes(dg[0], R, 2 * G4, True, G4, x2=dg[1], seg=G4, axis=0)
    wt = _planes(wf, D, 2 * G4, False, D, x2=wr, seg=G4, axis=0)
    dx = torch.empty(R, D, device=dev)
    ops.gemm_x6(da, R, 2 * G4, wt, D, 2 * G4, R, D, 2 * G4, dx, D)
    ex = torch.cat([dg[0], dg[1]], 1).double() @ w.double()
    assert float((dx.double() - ex).abs().max()) <= 3e-6 * float(ex.abs().max())
    # weight gradients: transposed plane set of both gate planes along the rows, outputs cut along the rows; the recurrent product as k windows
    dgt = _planes(dg[0], 2 * G4, R, False, G4, x2=dg[1], seg=G4, axis=1)
    xt = _planes(x, D, R, False, D)
    dw = torch.empty(2, G4, D, device=dev)
    ops.gemm_x6(dgt, 2 * G4, R, xt, D, R, 2 * G4, D, R, dw[0], D, c1=dw[1], rsplit=G4)
    ex = torch.cat([dg[0], dg[1]], 1).double().T @ x.double()
    assert float((dw.reshape(2 * G4, D).double() - ex).abs().max()) <= 3e-6 * float(ex.abs().max())
    yt = _planes(y, 2 * Hh, R, False, 2 * Hh)
    dwh = torch.empty(2, G4, Hh, device=dev)
    ops.gemm_x6(dgt, 2 * G4, R, yt, 2 * Hh, R, G4, Hh, R - sh, dwh[0], Hh, a_row0=0, a_kk0=sh // 16, b_row0=0, b_kk0=0)
    ops.gemm_x6(dgt, 2 * G4, R, yt, 2 * Hh, R, G4, Hh, R - sh, dwh[1], Hh, a_row0=G4, a_kk0=0, b_row0=Hh, b_kk0=sh // 16)
    ex0 = dg[0][sh:].double().T @ y[:R - sh, :Hh].double()
    ex1 = dg[1][:R - sh].double().T @ y[sh:, Hh:].double()
    assert float((dwh[0].double() - ex0).abs().max()) <= 3e-6 * float(ex0.abs().max())
    assert float((dwh[1].double() - ex1).abs().max()) <= 3e-6 * float(ex1.abs().max())
    # the same two products as ONE launch with two view sets (rows >= G4 of the stacked result: the reverse direction's views)
    dw2 = torch.full((2, G4, Hh), float("nan"), device=dev)
    ops.gemm_x6_two_views(dgt, 2 * G4, R, yt, 2 * Hh, R, 2 * G4, Hh, R - sh, G4, (sh // 16, 0, 0), (0, Hh, sh // 16), dw2[0], dw2[1], Hh)
    assert float((dw2[0].double() - ex0).abs().max()) <= 3e-6 * float(ex0.abs().max())
    assert float((dw2[1].double() - ex1).abs().max()) <= 3e-6 * float(ex1.abs().max())


_LAYER_CODE = r'''
import sys, torch
sys.path.insert(0, %(root)r)
from vistaocr_amd import ops
T, B, D, H, packed = %(T)d, %(B)d, %(D)d, 512, %(packed)d
g = torch.Generator().manual_seed(4)
dev = torch.device("cuda:0")
lens = sorted([max(1, T - (3 * i) // 2) for i in range(B)], reverse=True)
x = ((torch.rand(T * B, D, generator=g) - 0.5) * 2).to(dev)
params = [((torch.rand(*s, generator=g) - 0.5) * 0.16).to(dev).requires_grad_(True) for s in [(4 * H, D), (4 * H, H), (4 * H,), (4 * H,)] * 2]
lens_dev = torch.tensor(lens, dtype=torch.int32, device=dev)
rows = 0
if packed:
    maps = ops.SeqRowMaps(lens_dev, lens, T, B)
    x = ops.gather_rows(x, maps.to_dense, maps.rows)
    rows = maps.rows
xg = x.clone().requires_grad_(True)
y = ops.BiLstmLayerFn.apply(xg, lens_dev, T, B, *params, None, False, 0.0, 0, rows)
dy = ((torch.rand(y.shape, generator=g) - 0.5) * 0.1).to(dev)
if packed:
    dy = dy * (maps.to_dense >= 0).float().unsqueeze(1)
y.backward(dy)
torch.save({"y": y.detach().cpu(), "dx": xg.grad.cpu(), "g": [p.grad.cpu() for p in params]}, sys.argv[1])
'''


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("T,B,D,packed", [(24, 32, 1024, 0), (30, 27, 1024, 0), (40, 32, 1024, 1), (20, 32, 128, 0)])
def test_bilstm_layer_with_and_without_the_split_products(dev, tmp_path, T, B, D, packed, scheme):
    outs = []
    for mode in (scheme, "f32"):
        f = str(tmp_path / ("l%s.pt" % mode))
        r = subprocess.run([sys.executable, "-c", _LAYER_CODE % dict(root=ROOT, T=T, B=B, D=D, packed=packed), f],
                           env=dict(os.environ, VOCR_LSTM_GEMM=mode), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(torch.load(f))
    a, b = outs
    assert float((a["y"] - b["y"]).abs().max()) <= 2e-5
    assert float((a["dx"] - b["dx"]).abs().max()) <= 1e-4 * max(1e-3, float(b["dx"].abs().max()))
    for ga, gb in zip(a["g"], b["g"]):
        assert float((ga - gb).abs().max()) <= 2e-4 * max(1e-3, float(gb.abs().max()))


def test_the_model_in_fp16x3_mode_holds_the_oracle_bars():
    """VOCR_LSTM_GEMM=fp16x3 (the opt-in split of the LSTM's large GEMMs) under the SAME full-size oracle tests the default mode answers to:
    configs[1] at B = 32 (labels bit-exact, loss, every gradient tensor element-wise) and configs[3] at B = 32 on packed rows."""
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu",
                        os.path.join(ROOT, "tests", "test_round2_gpu.py::test_config1_full_size_vs_oracle"),
                        os.path.join(ROOT, "tests", "test_configs_gpu.py::test_config4_at_batch_size_32_hidden512")],
                       env=dict(os.environ, VOCR_LSTM_GEMM="fp16x3"), capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1500:])
    assert "2 passed" in r.stdout, r.stdout[-500:]
