"""GPU: the remaining BASELINE.json configs against the on-box oracle (same seeded inputs, sizes the CPU oracle
finishes in seconds): config 4 — MADCAT-style Arabic alphabet (V=166), wide lines up to 1200 px, variable widths in
SortByWidthCollater layout; config 5 — 60-px lines through rapid_ds (60 -> 30) with the 512-hidden BiLSTM.
Integer outputs (lens, greedy label sequences) bit-exact - the batch seed is chosen on the oracle side so that no frame's decision is
within the decode margin of flipping - CTC loss 1e-3, every gradient tensor element-wise 1e-2."""
import numpy as np
import pytest
import torch

from oracle import closed_form as cf
from oracle import vista_oracle as vo

pytestmark = pytest.mark.gpu


def _masks(T, B, H, n, seed):
    r = np.random.RandomState(seed)
    return [torch.from_numpy((r.uniform(size=(T, B, 2 * H)) >= 0.5).astype(np.float32) * 2.0) for _ in range(n)]


def _hp(h_img=30, rds=30, din=32, layers=1, hidden=32, drop=0.0, cin=1, **kw):
    return dict(input_line_height=h_img, rds_line_height=rds, lstm_input_dim=din, num_lstm_layers=layers, num_lstm_hidden_units=hidden,
                p_lstm_dropout=drop, num_in_channels=cin, **kw)


# name -> arguments of _run_pair.  `seed` is the first batch seed tried; scripts/margin_search.py cases runs the oracle-side search of
# every case on the CPU and prints the seed it settles on, so the seeds below make the first try succeed.
CASES = {
    "config4_arabic": dict(hp=_hp(din=64, layers=2, hidden=64, drop=0.5), alphabet="arabic", B=6, widths=[1200, 1113, 907, 640, 333, 15],
                           labels_per_line=[40, 33, 25, 17, 9, 1], seed=8, ltr=False),
    # config 4 on the DEFAULT sweep kernels: H = 512, 3 layers, B = 8 (one 8-row chain per direction), ragged T = 588 ... 7, dropout masks
    "config4_h512": dict(hp=_hp(din=128, layers=3, hidden=512, drop=0.5), alphabet="arabic", B=8,
                         widths=[1200, 1113, 907, 640, 333, 150, 64, 15], labels_per_line=[40, 33, 25, 17, 9, 4, 2, 1], seed=6, ltr=False,
                         state_kw=dict(lstm_scale=0.08, prob_scale=2.0), masks=(588, 8, 512, 2, 44)),
    # the 0.3-scaled closed form saturates the 512-unit gates (|logit| ~ 20, logit error ~ 3e-3): ask for a wider margin
    "config5_rds": dict(hp=_hp(h_img=60, din=128, layers=3, hidden=512, drop=0.5), alphabet="english", B=3, widths=[400, 322, 128],
                        labels_per_line=[12, 9, 3], seed=15, want_margin=1e-2, tries=200),
    "config5_fp16": dict(hp=_hp(h_img=60, din=128, layers=3, hidden=512, drop=0.5, conv_dtype="fp16"), alphabet="english", B=3,
                         widths=[400, 322, 128], labels_per_line=[12, 9, 3], seed=11, loss_rtol=1e-2, logit_rtol=2e-2, grad_rtol=0.1,
                         min_label_agreement=0.97, state_kw=dict(lstm_scale=0.08, prob_scale=0.5)),
    # config 4 at BATCH SIZE (round 4): B = 32, H = 512, 3 layers, the ragged MADCAT-distributed widths of bench.py --config c4 (src/madcat.py:58-66
    # scaled to <= 1200 px: 1178 ... 131, mean 662), W // 30 labels per line, dropout masks - the 4-row chain sweeps run every chain to max(len)
    "config4_b32": dict(hp=_hp(din=128, layers=3, hidden=512, drop=0.5), alphabet="arabic", B=32,
                        widths=[1178, 1154, 1142, 1123, 1089, 1063, 1025, 866, 863, 786, 779, 733, 715, 699, 660, 659, 639, 631, 589, 588, 567, 549,
                                442, 413, 398, 373, 368, 364, 255, 199, 136, 131],
                        labels_per_line=[39, 38, 38, 37, 36, 35, 34, 28, 28, 26, 25, 24, 23, 23, 22, 21, 21, 21, 19, 19, 18, 18, 14, 13, 13, 12, 12,
                                         12, 8, 6, 4, 4], seed=23, ltr=False, want_margin=6e-4, tries=60,      # measured logit error 1.6e-4 (bf16x6 x-projections on packed rows)
                        state_kw=dict(lstm_scale=0.08, prob_scale=2.0), masks=(576, 32, 512, 2, 46)),
    # config 5 at BATCH SIZE (round 6): bench.py --config c5's shape - 32 x 1x60x1200, rapid_ds 60 -> 30, fp16 conv operands, 3 x BiLSTM-512,
    # T = 294 - forward AND backward; the fp16 forward / data-gradient launches of this shape take the tile variants 64x2, 128x2, 128x4
    # and 128x4c2 (vocr_conv3x3_h16_plan), which the B = 3 case above does not
    "config5_b32": dict(hp=_hp(h_img=60, din=128, layers=3, hidden=512, drop=0.5, conv_dtype="fp16"), alphabet="english", B=32,
                        widths=[1200] * 32, labels_per_line=[40] * 32, seed=1, loss_rtol=1e-2, logit_rtol=2e-2, grad_rtol=0.1,
                        min_label_agreement=0.97, state_kw=dict(lstm_scale=0.08, prob_scale=0.5), masks=(294, 32, 512, 2, 48)),
    # the reference's own speed_test.py shape (src/speed_test.py:16-30): batch 64, height 30, lstm_input_dim 128, 3 x BiLSTM-512, dropout 0.5,
    # 5 labels per line, one of its widths (300); the LSTM sweeps run as two 32-row batch tiles
    "speed_test_b64": dict(hp=_hp(din=128, layers=3, hidden=512, drop=0.5), alphabet="english", B=64, widths=[300] * 64, labels_per_line=[5] * 64,
                           seed=2, want_margin=3e-4, tries=60, state_kw=dict(lstm_scale=0.08, prob_scale=2.0), masks=(147, 64, 512, 2, 47)),
    "rgb": dict(hp=_hp(cin=3), alphabet="english", B=2, widths=[150, 90], labels_per_line=[5, 3], seed=3),
    "batch40": dict(hp=_hp(hidden=64), alphabet="english", B=40, widths=[100] * 30 + [64] * 10, labels_per_line=[3] * 40, seed=5),
    # more than 64 lines per batch (the reference's --batch-size is free, src/train_cnn_lstm.py:155): the LSTM runs as two batch tiles, the
    # second one over its own shorter T; 2 layers so that a tiled layer feeds a tiled layer; H = 256 = the 8-row chain kernels (B tile 48 > 32:
    # 16-row chains) - and dropout masks
    "batch96": dict(hp=_hp(din=64, layers=2, hidden=256, drop=0.5), alphabet="english", B=96, widths=[120] * 40 + [90] * 30 + [64] * 26,
                    labels_per_line=[3] * 96, seed=5, state_kw=dict(lstm_scale=0.08, prob_scale=2.0), masks=(58, 96, 256, 1, 45)),
}


def _inputs(case, seed):
    from tests import golden_util as gu
    chars = gu.alphabet_chars(case["alphabet"])
    hp, B = case["hp"], case["B"]
    x, w, tgt, tl = cf.closed_form_batch(B, hp.get("num_in_channels", 1), hp["input_line_height"], case["widths"], len(chars),
                                         case["labels_per_line"], seed=seed)
    r = np.random.RandomState(seed + 100)
    s1 = torch.from_numpy(r.uniform(0, 0.999, size=(B, 64, 2)).astype(np.float32))
    s2 = torch.from_numpy(r.uniform(0, 0.999, size=(B, 128, 2)).astype(np.float32))
    return chars, x, w, tgt, tl, s1, s2


def _oracle_kw(masks):
    return dict(training=True, dropout_masks=masks) if masks is not None else dict(training=True, lstm_training=False)


def pick_case_seed(case, sd_np=None, masks=None):
    """Oracle side only (no GPU): the first batch seed >= case['seed'] whose greedy decode is `want_margin` away from flipping."""
    from tests import golden_util as gu
    from tests import parity_util as pu
    V = len(gu.alphabet_chars(case["alphabet"]))
    if sd_np is None:
        sd_np = cf.closed_form_state(case["hp"], V, **(case.get("state_kw") or {}))
    if masks is None and case.get("masks"):
        masks = _masks(*case["masks"])

    def oracle_logits(seed):
        _, x, w, _, _, s1, s2 = _inputs(case, seed)
        with torch.no_grad():
            lo, ln = vo.forward(vo.state_from_numpy(sd_np, requires_grad=False), case["hp"], torch.from_numpy(x), w, (s1, s2), **_oracle_kw(masks))
        return lo, ln, V

    seed, margin, _ = pu.pick_seed(oracle_logits, case["seed"], want=case.get("want_margin", 1e-3), tries=case.get("tries", 40))
    return seed, margin


def _run_pair(name):
    """One train-mode forward + backward on the HIP path and on the oracle.  For the fp32 path (min_label_agreement == 1) the batch
    seed is advanced on the ORACLE side until its greedy decode is at least `want_margin` away from flipping on every valid frame;
    the label sequences are then compared bit for bit, unconditionally.  Gradients: every tensor element-wise."""
    import os
    import vistaocr_amd as va
    from tests import parity_util as pu
    case = CASES[name]
    hp = case["hp"]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    masks = _masks(*case["masks"]) if case.get("masks") else None
    exact = case.get("min_label_agreement", 1.0) >= 1.0
    seed, margin = case["seed"], None
    chars = _inputs(case, seed)[0]
    V = len(chars)
    sd_np = cf.closed_form_state(hp, V, **(case.get("state_kw") or {}))
    if exact:
        seed, margin = pick_case_seed(case, sd_np, masks)
    chars, x, w, tgt, tl, s1, s2 = _inputs(case, seed)
    al = va.Alphabet(chars, left_to_right=case.get("ltr", True))
    model = va.CnnOcrModel(alphabet=al, verbose=False, **hp)
    sd = model.state_dict()
    for k, v in sd_np.items():
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    model.train()
    if masks is not None:
        model.dropout_masks = masks
    else:
        model.lstm.eval()
    model.pool_samples = [s1, s2]
    logits, lens = model(torch.from_numpy(x), torch.from_numpy(w))
    loss = va.CTCLoss()(logits, torch.from_numpy(tgt), lens, torch.from_numpy(tl))
    loss.backward()
    osd = vo.state_from_numpy(sd_np)
    lo, ln = vo.forward(osd, hp, torch.from_numpy(x), w, (s1, s2), **_oracle_kw(masks))
    lo_loss = vo.ctc_criterion(lo, torch.from_numpy(tgt), ln, torch.from_numpy(tl))
    lo_loss.backward()
    assert lens.tolist() == ln.tolist()
    rel = abs(float(loss.detach()) - float(lo_loss.detach())) / abs(float(lo_loss.detach()))
    assert rel <= case.get("loss_rtol", 1e-3), (float(loss.detach()), float(lo_loss.detach()))
    lg = logits.detach().cpu()
    T = lg.shape[0]
    valid = torch.arange(T).unsqueeze(1) < ln.unsqueeze(0)
    # fp32 logits with a different summation order: tolerance relative to the logit scale (the 512-hidden, 3-layer
    # closed-form model saturates its gates and reaches |logit| ~ 20, measured error 3e-3 there, 2e-5 at H=48)
    scale = max(1.0, float(lo.detach().abs()[valid].max()))
    err = float((lg - lo.detach()).abs()[valid].max())
    assert err < case.get("logit_rtol", 5e-4) * scale, (err, scale)
    emitted = -1
    if exact:
        olabels = vo.greedy_decode(lo.detach(), ln, al.idx_to_char, uxxxx=True)[1]
        emitted = sum(len(l) for l in olabels)
        assert err <= margin / 2, "logit error %.2e against an oracle decode margin of %.2e" % (err, margin)
        assert emitted >= 1, "no label emitted: the comparison would be vacuous"
        assert model.decode_labels(logits, lens) == olabels, "greedy label sequences differ"
    else:       # fp16 conv operands (config 5): agreement rate on well-separated frames, SURVEY.md §7
        top2 = torch.sort(lo.detach(), dim=2, descending=True)[0]
        safe = valid & ((top2[:, :, 0] - top2[:, :, 1]) > 1e-3)
        agree = float((lg.argmax(2)[safe] == lo.detach().argmax(2)[safe]).float().mean())
        assert agree >= case["min_label_agreement"], "per-frame argmax agreement %.4f on well-separated frames" % agree
    worst = pu.assert_grads_close(model, osd, rtol=case.get("grad_rtol"))
    print("%s (batch seed %d): loss rel %.2e, max |dlogit| %.2e at scale %.1f, oracle decode margin %s, labels emitted %d, worst "
          "element-wise gradient error %.2e (%s)" % (name, seed, rel, err, scale, "%.2e" % margin if margin else "-", emitted, worst[1], worst[0]))
    return model, logits, lens


def test_config4_arabic_wide_variable_width():
    model, logits, lens = _run_pair("config4_arabic")
    assert lens.tolist() == [588, 545, 443, 313, 163, 7]
    assert logits.shape[2] == 166


def test_config4_wide_ragged_hidden512_on_the_8row_sweeps():
    """Config 4 on the DEFAULT sweep kernels: H = 512, 3 layers, B = 8 (one 8-row chain per direction), widths 1200 ... 15, i.e. ragged
    T = 588 ... 7 through lstm_fwd_chain8 / lstm_bwd_kowner8 with explicit inter-layer dropout masks, Arabic alphabet (V = 166)."""
    model, logits, lens = _run_pair("config4_h512")
    assert lens.tolist() == [588, 545, 443, 313, 163, 73, 30, 7]
    assert logits.shape[2] == 166


def test_config5_rds_60px_hidden512():
    model, logits, lens = _run_pair("config5_rds")
    assert lens.tolist() == [98, 78, 30]


def test_config5_fp16_conv_mfma():
    """Config 5 as BASELINE.json words it: 60-px lines, rapid_ds, 512-hidden BiLSTM, fp16 conv MFMA with fp32 accumulate
    (CTC in fp32).  The oracle applies the same operand rounding (oracle/vista_oracle.py _ConvF16Operands)."""
    # fp16 operand rounding is a discontinuity: a 1e-7 difference in an activation (summation order) can flip its
    # rounding at the next layer, which acts as 2^-11-sized noise that the saturated 512-hidden closed-form LSTM
    # amplifies.  Per SURVEY.md §7, fp16 results get a looser loss tolerance and a label-agreement rate; the
    # 1e-3 / bit-exact contract applies to the fp32 path only.
    # (recurrent weights drawn from the reference's own init range +-0.08 so the 512-hidden LSTM is not saturated)
    model, logits, lens = _run_pair("config5_fp16")
    assert lens.tolist() == [98, 78, 30] and model.conv_dtype == "fp16"


def test_rgb_input_three_channels():
    _run_pair("rgb")


def test_batch_96_runs_as_lstm_batch_tiles():
    """B = 96 > 64: CnnOcrModel splits every BiLSTM layer into batch tiles of <= 64 rows (model._bilstm_layer)."""
    model, logits, lens = _run_pair("batch96")
    assert logits.shape[1] == 96 and lens.tolist() == [58] * 40 + [43] * 30 + [30] * 26


def test_batch_40_is_one_lstm_tile_of_ragged_rows():
    """B = 40 at H = 64: one batch tile with more rows than a 32-row chain set."""
    _run_pair("batch40")


def test_batch_64_speed_test_shape():
    """The reference's own speed_test.py shape (src/speed_test.py:16-30): batch 64, 3 x BiLSTM-512, lstm_input_dim 128, 5 labels per
    line, width 300 - one train step against the oracle."""
    model, logits, lens = _run_pair("speed_test_b64")
    assert logits.shape[1] == 64 and lens.tolist() == [147] * 64


def test_config4_at_batch_size_32_hidden512():
    """BASELINE configs[3] at batch size: V = 166, B = 32, H = 512, widths from the MADCAT distribution scaled to <= 1200 px (the batch of
    bench.py --config c4), ragged T = 576 ... 63, dropout masks; loss, greedy labels (bit-exact), logits and every gradient tensor
    against the oracle."""
    import bench
    assert CASES["config4_b32"]["widths"] == bench.WORKLOADS["c4"]["widths"] and CASES["config4_b32"]["labels_per_line"] == bench.WORKLOADS["c4"]["labels"]
    model, logits, lens = _run_pair("config4_b32")
    assert logits.shape[1:] == (32, 166) and lens[0] == logits.shape[0] and lens.tolist() == sorted(lens.tolist(), reverse=True)


def test_config5_at_batch_size_32():
    """BASELINE configs[4] at the size bench.py --config c5 runs, with a backward pass: loss <= 1e-2, per-frame agreement >= 0.97 on safe
    frames, every gradient at the fp16 bar - and every fp16 forward / data-gradient tile variant that shape takes is one GPUTEST sees."""
    import math
    import bench
    from vistaocr_amd import _lib
    case = CASES["config5_b32"]
    wl = bench.WORKLOADS["c5"]
    assert case["widths"] == wl["widths"] and case["labels_per_line"] == wl["labels"] and wl["conv_dtype"] == "fp16" and wl["himg"] == 60
    lib = _lib.load()
    h, w, seen = 30, 600, set()
    for st in [(16, 64), (64, 64), "pool", (64, 128), (128, 128), "pool", (128, 256), (256, 256), (256, 256)]:
        if st == "pool":
            h, w = math.floor(h * 0.5), math.floor(w * 0.7)
            continue
        seen.add(lib.vocr_conv3x3_h16_plan(32, st[0], h, w, st[1]))            # forward
        seen.add(lib.vocr_conv3x3_h16_plan(32, st[1], h, w, st[0]))            # data gradient
    assert seen == {1, 2, 4, 5}, seen                    # 64x2, 128x2, 128x4, 128x4c2 (128x3: other batch sizes)
    model, logits, lens = _run_pair("config5_b32")
    assert logits.shape[0] == 294 and lens.tolist() == [294] * 32


def test_validation_pass_and_snapshot_roundtrip(tmp_path):
    """test_on_val (src/train_cnn_lstm.py:32-100) on the HIP path + checkpoint written/read with the reference schema."""
    import os
    import vistaocr_amd as va
    from vistaocr_amd.loop import SortByWidthCollater, save_snapshot, test_on_val
    al = va.english_alphabet()
    hp = dict(input_line_height=30, rds_line_height=30, lstm_input_dim=32, num_lstm_layers=1, num_lstm_hidden_units=32,
              p_lstm_dropout=0.0, num_in_channels=1)
    torch.manual_seed(1)
    model = va.CnnOcrModel(alphabet=al, verbose=False, **hp)
    r = np.random.RandomState(0)
    items = []
    for i, w in enumerate([120, 90, 150, 64, 200, 33]):
        items.append((torch.from_numpy(r.uniform(0, 1, size=(1, 30, w)).astype(np.float32)), [int(v) for v in r.randint(1, 96, size=4)],
                      {"width": w, "utt-id": "u%d" % i}))
    loader = [SortByWidthCollater(items[:3]), SortByWidthCollater(items[3:])]
    torch.manual_seed(7)
    loss, cer, wer = test_on_val(loader, model, va.CTCLoss())
    assert np.isfinite(loss) and 0.0 <= cer and 0.0 <= wer and model.training
    opt = va.make_optimizer(model)
    path = os.path.join(tmp_path, "snap.pth")
    save_snapshot(path, 3, model, opt, False, 1e-3, loss, cer, wer, 30)
    m2 = va.CnnOcrModel.FromSavedWeights(path, verbose=False)
    for k, v in model.state_dict().items():
        assert torch.equal(v.cpu(), m2.state_dict()[k].cpu()), k


def test_decode_dataset_writes_reference_hyp_files(tmp_path):
    """decode_testset.py's loop + file format on the HIP path: line count, "<uxxxx ...> (<utt-id>)" / "<utf8> (<utt-id minus the last
    _part>)" framing, the two files agreeing through uxxxx_to_utf8.  (What the strings must BE is pinned elsewhere: the decode edge-case
    fixture and the golden label sequences.)"""
    import os
    import vistaocr_amd as va
    from vistaocr_amd.loop import SortByWidthCollater, decode_dataset
    al = va.english_alphabet()
    hp = dict(input_line_height=30, rds_line_height=30, lstm_input_dim=32, num_lstm_layers=1, num_lstm_hidden_units=32,
              p_lstm_dropout=0.0, num_in_channels=1)
    sd_np = cf.closed_form_state(hp, len(al))
    model = va.CnnOcrModel(alphabet=al, verbose=False, **hp)
    sd = model.state_dict()
    for k, v in sd_np.items():
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    r = np.random.RandomState(0)
    items = [(torch.from_numpy(r.uniform(0, 1, size=(1, 30, w)).astype(np.float32)), [1], {"width": w, "utt-id": "doc7_line_%d" % i})
             for i, w in enumerate([140, 96, 201, 64])]
    loader = [SortByWidthCollater(items[:2]), SortByWidthCollater(items[2:])]
    n = decode_dataset(model, loader, str(tmp_path))
    assert n == 4
    a = open(os.path.join(tmp_path, "hyp-chars.txt")).read().splitlines()
    b = open(os.path.join(tmp_path, "hyp-chars.txt.utf8")).read().splitlines()
    assert len(a) == len(b) == 4
    for la, lb in zip(a, b):
        ux, uid = la.rsplit(" (", 1)
        u8, uid8 = lb.rsplit(" (", 1)
        assert uid.rstrip(")").startswith("doc7_line_") and uid8.rstrip(")") == "doc7_line"
        assert vo.uxxxx_to_utf8(ux) == u8
        assert all(tok.startswith("u") and len(tok) == 5 for tok in ux.split()) or ux == ""
