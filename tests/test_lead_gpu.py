"""GPU: vocr_lstm_fwd_lead (include/vocr.h) - the forward sweep that takes its x-projection as two source-direction planes and, with
next_wpack, computes the NEXT layer's x-projection on four more waves of every workgroup (src/models/cnnlstm.py:148-149,288-290:
cuDNN's multi-layer RNN overlaps the two).
  * the sweep's own outputs (y, gates, cell) are bit-identical to vocr_lstm_fwd / vocr_lstm_fwd_packed, with and without followers;
  * the planes against an fp64 product of the sweep's own y (x the dropout mask), dense and packed rows, ragged lengths, a partial
    last chain; run to run bit-identical;
  * xproj = plane 0 + plane 1 read by the next sweep: the stack of two layers against the GEMM path;
  * the whole model with VOCR_LSTM_FOLLOW=1 against the default path: logits, loss, labels, every gradient (drawn dropout included:
    the mask is the same counter-based draw), also on packed rows;
  * GatherRowsFn's fill gradient (the bias of the output layer under a loss that is not zero on padded frames)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = 512
G = 4 * H


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rnd(g, dev, *s, a=1.0):
    return ((torch.rand(*s, generator=g) - 0.5) * a).to(dev)


def _p(t):
    return t.data_ptr() if t is not None else None


@pytest.mark.parametrize("T,B,masked,packed", [(37, 32, True, False), (50, 27, False, False), (20, 17, True, False), (41, 32, True, True),
                                               (33, 21, False, True)])
def test_follower_planes_and_sweep_outputs(dev, T, B, masked, packed):
    from vistaocr_amd import _lib, ops
    from vistaocr_amd._lib import call
    lib = _lib.load()
    assert lib.vocr_lstm_follow_supported(B, H) == 1
    g = torch.Generator().manual_seed(T * 100 + B)
    lens_l = sorted([max(1, T - (5 * i) // 2) for i in range(B)], reverse=True)
    lens = torch.tensor(lens_l, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    if packed:
        maps = ops.SeqRowMaps(lens, lens_l, T, B)
        R, rows = maps.rows, maps.rows
    else:
        maps, R, rows = None, T * B, 0
    xproj = _rnd(g, dev, 2, R, G, a=0.6)
    wf, wr = _rnd(g, dev, G, H, a=0.2), _rnd(g, dev, G, H, a=0.2)
    nwf, nwr = _rnd(g, dev, G, 2 * H, a=0.16), _rnd(g, dev, G, 2 * H, a=0.16)
    nb = _rnd(g, dev, 2, G, a=0.1)
    mask = ops.dropout_mask(R, 2 * H, 0.5, 77, dev) if masked else None
    health = torch.zeros(2, dtype=torch.int32, device=dev)
    ws = torch.zeros(lib.vocr_lstm_workspace_bytes(T, B, H) // 4 + 16, device=dev)
    mk = lambda: ((torch.zeros if packed else torch.empty)(R, 2 * H, device=dev), torch.zeros(2, R, G, device=dev), torch.zeros(2, R, H, device=dev))
    y0, g0, c0 = mk()
    if packed:
        call("vocr_lstm_fwd_packed", _p(xproj), _p(wf), _p(wr), _p(lens), _p(y0), _p(g0), _p(c0), _p(ws), T, B, H, R, _p(health), s)
    else:
        call("vocr_lstm_fwd", _p(xproj), _p(wf), _p(wr), _p(lens), _p(y0), _p(g0), _p(c0), _p(ws), T, B, H, _p(health), s)
    wpack = ops.lstm_xproj_pack(nwf, nwr)
    outs = []
    for run in range(2):
        y, gt, c = mk()
        planes = torch.full((2, 2, R, G), float("nan"), device=dev)
        call("vocr_lstm_fwd_lead", _p(xproj), None, _p(wf), _p(wr), _p(lens), _p(y), _p(gt), _p(c), _p(ws), T, B, H, rows, _p(wpack), _p(nb), _p(mask),
             _p(planes), _p(health), s)
        torch.cuda.synchronize()
        assert health.tolist() == [0, 0]
        outs.append((y, gt, c, planes))
    y, gt, c, planes = outs[0]
    live = (maps.to_dense >= 0) if packed else torch.ones(R, dtype=torch.bool, device=dev)
    assert torch.equal(y[live], y0[live]) and torch.equal(gt[:, live], g0[:, live]) and torch.equal(c[:, live], c0[:, live])
    assert torch.equal(outs[1][3][:, :, live], planes[:, :, live]), "planes differ run to run"
    # without followers the same entry point is the plain sweep
    y1, g1, c1 = mk()
    call("vocr_lstm_fwd_lead", _p(xproj), None, _p(wf), _p(wr), _p(lens), _p(y1), _p(g1), _p(c1), _p(ws), T, B, H, rows, None, None, None, None, _p(health), s)
    assert torch.equal(y1[live], y0[live]) and torch.equal(g1[:, live], g0[:, live])
    yy = (y * mask if masked else y).double()
    W = torch.cat([nwf, nwr], 0).double()
    for src in (0, 1):
        ref = yy[:, src * H:(src + 1) * H] @ W[:, src * H:(src + 1) * H].T
        if src == 0:
            ref = ref + nb.reshape(-1).double()
        got = torch.cat([planes[src, 0], planes[src, 1]], 1).double()
        d = (got[live] - ref[live]).abs().max().item()
        assert not torch.isnan(got[live]).any() and d <= 2e-6 * max(1.0, ref.abs().max().item()) + 2e-6, (src, d)


def test_two_layers_through_the_planes_match_the_gemm_path(dev):
    """Layer 1 reads xproj = plane 0 + plane 1 written by layer 0's followers: against layer 1 on the x-projection GEMM of layer 0's y."""
    from vistaocr_amd import _lib, ops
    from vistaocr_amd._lib import call
    lib = _lib.load()
    T, B = 45, 32
    R = T * B
    g = torch.Generator().manual_seed(5)
    lens = torch.tensor(sorted([max(1, T - i) for i in range(B)], reverse=True), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    xproj = _rnd(g, dev, 2, R, G, a=0.6)
    w = [(_rnd(g, dev, G, H, a=0.2), _rnd(g, dev, G, H, a=0.2)) for _ in range(2)]
    nwf, nwr = _rnd(g, dev, G, 2 * H, a=0.16), _rnd(g, dev, G, 2 * H, a=0.16)
    nb = _rnd(g, dev, 2, G, a=0.1)
    health = torch.zeros(2, dtype=torch.int32, device=dev)
    ws = torch.zeros(lib.vocr_lstm_workspace_bytes(T, B, H) // 4 + 16, device=dev)
    mk = lambda: (torch.empty(R, 2 * H, device=dev), torch.empty(2, R, G, device=dev), torch.empty(2, R, H, device=dev))
    y0, g0, c0 = mk()
    planes = torch.empty(2, 2, R, G, device=dev)
    call("vocr_lstm_fwd_lead", _p(xproj), None, _p(w[0][0]), _p(w[0][1]), _p(lens), _p(y0), _p(g0), _p(c0), _p(ws), T, B, H, 0,
         _p(ops.lstm_xproj_pack(nwf, nwr)), _p(nb), None, _p(planes), _p(health), s)
    y1, g1, c1 = mk()
    call("vocr_lstm_fwd_lead", _p(planes[0]), _p(planes[1]), _p(w[1][0]), _p(w[1][1]), _p(lens), _p(y1), _p(g1), _p(c1), _p(ws), T, B, H, 0,
         None, None, None, None, _p(health), s)
    xo = torch.empty(2, R, G, device=dev)
    ops.gemm_pair(0, 0, 1, R, G, 2 * H, y0, y0, 2 * H, nwf, nwr, 2 * H, xo[0], xo[1], G, bias0=nb[0], bias1=nb[1])
    y2, g2, c2 = mk()
    call("vocr_lstm_fwd", _p(xo), _p(w[1][0]), _p(w[1][1]), _p(lens), _p(y2), _p(g2), _p(c2), _p(ws), T, B, H, _p(health), s)
    torch.cuda.synchronize()
    assert health.tolist() == [0, 0]
    assert (y1 - y2).abs().max().item() <= 2e-5 and (g1 - g2).abs().max().item() <= 2e-5


_MODEL_CODE = r'''
import sys, torch
sys.path.insert(0, %(root)r)
import vistaocr_amd as va
from vistaocr_amd import ops
torch.manual_seed(3)
B, Hh, W = %(B)d, 30, %(W)d
m = va.CnnOcrModel(num_in_channels=1, input_line_height=Hh, rds_line_height=Hh, lstm_input_dim=128, num_lstm_layers=3, num_lstm_hidden_units=512,
                   p_lstm_dropout=0.5, alphabet=va.english_alphabet(), multigpu=False, verbose=False, gpu=True)
m.train()
g = torch.Generator().manual_seed(11)
x = torch.rand(B, 1, Hh, W, generator=g)
widths = torch.tensor(sorted([int(W - (%(ragged)d * i * W) // (2 * B)) for i in range(B)], reverse=True))
m.pool_samples = [torch.rand(B, 64, 2, generator=g), torch.rand(B, 128, 2, generator=g)]
m.pack_threshold = %(thr)f
logits, lens = m(x.cuda(), widths)
T = logits.shape[0]
wgt = torch.rand(logits.shape, generator=g).cuda()
loss = (logits * wgt).sum() + (logits ** 2).sum() * 0.01          # NOT zero on padded frames: the fill row's gradient is live
loss.backward()
ops.check_health_sync(torch.device("cuda:0"))
torch.save({"logits": logits.detach().cpu(), "lens": lens, "grads": {k: p.grad.detach().cpu() for k, p in m.named_parameters()}}, sys.argv[1])
'''


@pytest.mark.parametrize("ragged,thr", [(0, 0.0), (1, 0.0), (1, 0.95)])
def test_model_with_followers_matches_the_default_path(dev, tmp_path, ragged, thr):
    """VOCR_LSTM_FOLLOW=1 (experiment switch, off by default: measured slower) against the default path on a uniform batch,
    a ragged dense one and a ragged packed one, training mode with DRAWN inter-layer dropout."""
    outs = []
    for follow in ("0", "1"):
        f = str(tmp_path / ("o%s.pt" % follow))
        code = _MODEL_CODE % dict(root=ROOT, B=32, W=240, ragged=ragged, thr=thr)
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, VOCR_EXPERIMENTS="1", VOCR_LSTM_FOLLOW=follow), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(torch.load(f))
    a, b = outs
    assert a["lens"].tolist() == b["lens"].tolist()
    scale = a["logits"].abs().max().item()
    assert (a["logits"] - b["logits"]).abs().max().item() <= 2e-5 * max(1.0, scale)
    for k in a["grads"]:
        ga, gb = a["grads"][k], b["grads"][k]
        tol = 2e-4 * max(1e-3, ga.abs().max().item())
        assert (ga - gb).abs().max().item() <= tol, (k, (ga - gb).abs().max().item(), tol)


def test_fill_row_gradient_of_the_packed_path(dev, tmp_path):
    """The output layer's bias gradient under a loss that is not zero on padded frames: packed rows (the bias row fills the frames
    without a packed row: GatherRowsFn returns their column sum) against the dense path."""
    outs = []
    for thr in (0.0, 0.95):
        f = str(tmp_path / ("p%d.pt" % int(thr * 100)))
        code = _MODEL_CODE % dict(root=ROOT, B=12, W=200, ragged=1, thr=thr)
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(torch.load(f))
    a, b = outs
    k = [n for n in a["grads"] if n.startswith("prob_layer") and n.endswith("bias")][0]
    ga, gb = a["grads"][k], b["grads"][k]
    assert (ga - gb).abs().max().item() <= 1e-4 * max(1.0, ga.abs().max().item()), (ga - gb).abs().max().item()
    for n in a["grads"]:
        tol = 3e-4 * max(1e-3, a["grads"][n].abs().max().item())
        assert (a["grads"][n] - b["grads"][n]).abs().max().item() <= tol, n

