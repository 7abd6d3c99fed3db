"""GPU: the remaining BASELINE.json configs against the on-box oracle (same seeded inputs, sizes the CPU oracle
finishes in seconds): config 4 — MADCAT-style Arabic alphabet (V=166), wide lines up to 1200 px, variable widths in
SortByWidthCollater layout; config 5 — 60-px lines through rapid_ds (60 -> 30) with the 512-hidden BiLSTM.
Integer outputs (lens, labels on frames whose oracle top-2 margin exceeds fp32 noise) bit-exact; CTC loss 1e-3."""
import numpy as np
import pytest
import torch

from oracle import closed_form as cf
from oracle import vista_oracle as vo

pytestmark = pytest.mark.gpu


def _run_pair(hp, chars, B, widths, labels_per_line, seed, ltr=True, loss_rtol=1e-3, logit_rtol=5e-4, grad_rtol=1e-2,
              min_label_agreement=1.0, state_kw=None):
    import vistaocr_amd as va
    V = len(chars)
    sd_np = cf.closed_form_state(hp, V, **(state_kw or {}))
    x, w, tgt, tl = cf.closed_form_batch(B, hp.get("num_in_channels", 1), hp["input_line_height"], widths, V, labels_per_line, seed=seed)
    r = np.random.RandomState(seed + 100)
    s1 = r.uniform(0, 0.999, size=(B, 64, 2)).astype(np.float32)
    s2 = r.uniform(0, 0.999, size=(B, 128, 2)).astype(np.float32)
    al = va.Alphabet(chars, left_to_right=ltr)
    model = va.CnnOcrModel(alphabet=al, verbose=False, **hp)
    sd = model.state_dict()
    for k, v in sd_np.items():
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    model.train()
    model.lstm.eval()
    model.pool_samples = [torch.from_numpy(s1), torch.from_numpy(s2)]
    logits, lens = model(torch.from_numpy(x), torch.from_numpy(w))
    loss = va.CTCLoss()(logits, torch.from_numpy(tgt), lens, torch.from_numpy(tl))
    loss.backward()
    osd = vo.state_from_numpy(sd_np)
    lo, ln = vo.forward(osd, hp, torch.from_numpy(x), w, (torch.from_numpy(s1), torch.from_numpy(s2)), training=True, lstm_training=False)
    lo_loss = vo.ctc_criterion(lo, torch.from_numpy(tgt), ln, torch.from_numpy(tl))
    lo_loss.backward()
    assert lens.tolist() == ln.tolist()
    assert abs(float(loss) - float(lo_loss)) <= loss_rtol * abs(float(lo_loss)), (float(loss), float(lo_loss))
    lg = logits.detach().cpu()
    T = lg.shape[0]
    valid = torch.arange(T).unsqueeze(1) < ln.unsqueeze(0)
    # fp32 logits with a different summation order: tolerance relative to the logit scale (the 512-hidden, 3-layer
    # closed-form model saturates its gates and reaches |logit| ~ 20, measured error 3e-3 there, 2e-5 at H=48)
    scale = max(1.0, float(lo.detach().abs()[valid].max()))
    assert float((lg - lo.detach()).abs()[valid].max()) < logit_rtol * scale, (float((lg - lo.detach()).abs()[valid].max()), scale)
    top2 = torch.sort(lo.detach(), dim=2, descending=True)[0]
    margin = top2[:, :, 0] - top2[:, :, 1]
    safe = valid & (margin > 1e-3)
    agree = float((lg.argmax(2)[safe] == lo.detach().argmax(2)[safe]).float().mean())
    assert agree >= min_label_agreement, "per-frame argmax agreement %.4f on well-separated frames" % agree
    if min_label_agreement >= 1.0 and float(margin[valid].min()) > 1e-3:
        assert model.decode_labels(logits, lens) == vo.greedy_decode(lo.detach(), ln, al.idx_to_char, uxxxx=True)[1]
    for k, p in model.named_parameters():
        if k.startswith("cnn.") and k.endswith(".bias") and int(k.split(".")[1]) in (0, 3, 7, 10, 14, 17, 20):
            continue
        rn = float(osd[k].grad.double().norm())
        assert abs(float(p.grad.double().norm()) - rn) <= grad_rtol * rn + 1e-5, k
    return model, logits, lens


def test_config4_arabic_wide_variable_width():
    from tests import golden_util as gu
    chars = gu.alphabet_chars("arabic")
    hp = dict(input_line_height=30, rds_line_height=30, lstm_input_dim=64, num_lstm_layers=2, num_lstm_hidden_units=64,
              p_lstm_dropout=0.5, num_in_channels=1)
    widths = [1200, 1113, 907, 640, 333, 15]                  # MADCAT-like spread, sorted descending, min width 15
    model, logits, lens = _run_pair(hp, chars, 6, widths, [40, 33, 25, 17, 9, 1], seed=7, ltr=False)
    assert lens.tolist() == [588, 545, 443, 313, 163, 7]
    assert logits.shape[2] == 166


def test_config5_rds_60px_hidden512():
    from tests import golden_util as gu
    chars = gu.alphabet_chars("english")
    hp = dict(input_line_height=60, rds_line_height=30, lstm_input_dim=128, num_lstm_layers=3, num_lstm_hidden_units=512,
              p_lstm_dropout=0.5, num_in_channels=1)
    model, logits, lens = _run_pair(hp, chars, 3, [400, 322, 128], [12, 9, 3], seed=11)
    assert lens.tolist() == [98, 78, 30]


def test_config5_fp16_conv_mfma():
    """Config 5 as BASELINE.json words it: 60-px lines, rapid_ds, 512-hidden BiLSTM, fp16 conv MFMA with fp32 accumulate
    (CTC in fp32).  The oracle applies the same operand rounding (oracle/vista_oracle.py _ConvF16Operands)."""
    from tests import golden_util as gu
    chars = gu.alphabet_chars("english")
    hp = dict(input_line_height=60, rds_line_height=30, lstm_input_dim=128, num_lstm_layers=3, num_lstm_hidden_units=512,
              p_lstm_dropout=0.5, num_in_channels=1, conv_dtype="fp16")
    # fp16 operand rounding is a discontinuity: a 1e-7 difference in an activation (summation order) can flip its
    # rounding at the next layer, which acts as 2^-11-sized noise that the saturated 512-hidden closed-form LSTM
    # amplifies.  Per SURVEY.md §7, fp16 results get a looser loss tolerance and a label-agreement rate; the
    # 1e-3 / bit-exact contract applies to the fp32 path only.
    # (recurrent weights drawn from the reference's own init range +-0.08 so the 512-hidden LSTM is not saturated)
    model, logits, lens = _run_pair(hp, chars, 3, [400, 322, 128], [12, 9, 3], seed=11, loss_rtol=1e-2, logit_rtol=2e-2,
                                    grad_rtol=0.1, min_label_agreement=0.97, state_kw=dict(lstm_scale=0.08, prob_scale=0.5))
    assert lens.tolist() == [98, 78, 30] and model.conv_dtype == "fp16"


def test_rgb_input_three_channels():
    from tests import golden_util as gu
    chars = gu.alphabet_chars("english")
    hp = dict(input_line_height=30, rds_line_height=30, lstm_input_dim=32, num_lstm_layers=1, num_lstm_hidden_units=32,
              p_lstm_dropout=0.0, num_in_channels=3)
    _run_pair(hp, chars, 2, [150, 90], [5, 3], seed=3)


def test_batch_64_speed_test_shape():
    """The reference's own speed_test.py shape: batch 64 (src/speed_test.py:16): 4 MFMA row tiles in the sweeps."""
    import vistaocr_amd as va
    al = va.english_alphabet()
    hp = dict(input_line_height=30, rds_line_height=30, lstm_input_dim=32, num_lstm_layers=1, num_lstm_hidden_units=64,
              p_lstm_dropout=0.0, num_in_channels=1)
    from tests import golden_util as gu
    _run_pair(hp, gu.alphabet_chars("english"), 40, [100] * 30 + [64] * 10, [3] * 40, seed=5)


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
    """decode_testset.py's loop + file format on the HIP path; the strings equal the oracle's greedy decode of the same logits."""
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
