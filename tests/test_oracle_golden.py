"""CPU: pin the oracle restatement (oracle/vista_oracle.py) to the golden vectors that
oracle/gen_golden.py captured from the imported reference."""
import numpy as np
import pytest
import torch

from oracle import closed_form as cf
from oracle import vista_oracle as vo
from tests import golden_util as gu

CASES = [("c1", "english"), ("c1_eval", "english"), ("varwidth", "english"), ("varwidth_train", "english"),
         ("rds", "english"), ("arabic", "arabic")]


def _run(name, alpha):
    g, hp, chars, sd_np, x, w, tgt, tl, (s1, s2) = gu.case_inputs(name, alpha)
    sd = vo.state_from_numpy(sd_np)
    training = str(g["mode"]) == "train"
    taps = {}
    logits, lens = vo.forward(sd, hp, torch.from_numpy(x), w, (torch.from_numpy(s1), torch.from_numpy(s2)),
                              training=training, lstm_training=False, taps=taps)
    return g, hp, chars, sd, logits, lens, tgt, tl, taps


@pytest.mark.parametrize("name,alpha", CASES)
def test_forward_loss_labels_match_reference(name, alpha):
    g, hp, chars, sd, logits, lens, tgt, tl, taps = _run(name, alpha)
    assert lens.tolist() == g["lens"].tolist()
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=0, atol=2e-5)
    loss = vo.ctc_criterion(logits, torch.from_numpy(tgt), lens, torch.from_numpy(tl))
    assert loss.shape == (1,)
    np.testing.assert_allclose(loss.detach().numpy(), g["loss"], rtol=1e-6)
    strs, labels = vo.greedy_decode(logits.detach(), lens, dict(enumerate(chars)), uxxxx=True)
    assert strs == [str(s) for s in g["strings_uxxxx"]]
    assert labels == gu.split_labels(g)
    u8, _ = vo.greedy_decode(logits.detach(), lens, dict(enumerate(chars)), uxxxx=False)
    assert u8 == [str(s) for s in g["strings_utf8"]]
    if str(g["mode"]) == "train":
        loss.backward()
        for k, p in vo.trainable(sd):
            np.testing.assert_allclose(p.grad.double().norm().item(), float(g["gnorm/" + k]), rtol=2e-4, atol=1e-7)
            np.testing.assert_allclose(p.grad.reshape(-1)[:32].numpy(), g["ghead/" + k], rtol=2e-3, atol=2e-5)
        for k in g.files:
            if k.startswith("post/"):
                np.testing.assert_allclose(sd[k[5:]].numpy(), g[k], rtol=1e-5, atol=1e-6)


def test_explicit_lstm_matches_aten_lstm():
    """lstm_explicit (the explicit-mask contract) == nn.LSTM on packed variable-length input."""
    g, hp, chars, sd_np, x, w, tgt, tl, (s1, s2) = gu.case_inputs("varwidth_train")
    sd = vo.state_from_numpy(sd_np)
    T, B = 147, 4
    H, L = hp["num_lstm_hidden_units"], hp["num_lstm_layers"]
    ones = [torch.ones(T, B, 2 * H) for _ in range(L - 1)]
    a, la = vo.forward(sd, hp, torch.from_numpy(x), w, (torch.from_numpy(s1), torch.from_numpy(s2)), training=True,
                       lstm_training=False)
    b, lb = vo.forward(sd, hp, torch.from_numpy(x), w, (torch.from_numpy(s1), torch.from_numpy(s2)), training=True,
                       dropout_masks=ones)
    assert la.tolist() == lb.tolist()
    np.testing.assert_allclose(a.detach().numpy(), b.detach().numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(a.detach().numpy(), g["logits"], atol=2e-5, rtol=0)


def test_width_table_and_height():
    t = gu.load("width_table")
    hp30 = dict(input_line_height=30, rds_line_height=30)
    hp60 = dict(input_line_height=60, rds_line_height=30)
    for w, a, b in zip(t["widths"], t["t30"], t["t60"]):
        assert vo.output_size(hp30, 30, int(w))[1] == int(a)
        assert vo.output_size(hp60, 60, int(w))[1] == int(b)
    assert vo.output_size(hp30, 30, 20)[0] == int(t["h30"]) == 7
    assert vo.output_size(hp60, 60, 20)[0] == int(t["h60"])
    # SURVEY.md §8 a-6 spot values
    for w, T in [(15, 7), (150, 73), (200, 98), (300, 147), (350, 170), (450, 220), (599, 293), (600, 294), (601, 294),
                 (1200, 588)]:
        assert vo.output_size(hp30, 30, w)[1] == T


def test_fracpool_numpy_matches_aten():
    r = np.random.RandomState(0)
    x = r.uniform(-1, 1, size=(3, 5, 15, 47)).astype(np.float32)
    x[0, 0, 2:4, 5:9] = 0.5                                           # ties: first max must win
    u = r.uniform(0, 1, size=(3, 5, 2)).astype(np.float32)
    oh, ow = int(15 * 0.5), int(47 * 0.7)
    ref, idx = torch.nn.functional.fractional_max_pool2d(torch.from_numpy(x), 2, output_size=(oh, ow),
                                                         _random_samples=torch.from_numpy(u), return_indices=True)
    out, my_idx = vo.fracpool_numpy(x, u, oh, ow)
    np.testing.assert_array_equal(out, ref.numpy())
    np.testing.assert_array_equal(my_idx, idx.numpy())


def test_decode_edge_cases():
    g = gu.load("decode_edges")
    chars = gu.alphabet_chars("english")
    logits = torch.from_numpy(g["logits"])
    lens = torch.from_numpy(g["lens"])
    strs, labels = vo.greedy_decode(logits, lens, dict(enumerate(chars)), uxxxx=True)
    assert strs == [str(s) for s in g["strings_uxxxx"]]
    assert labels == gu.split_labels(g)
    u8, _ = vo.greedy_decode(logits, lens, dict(enumerate(chars)), uxxxx=False)
    assert u8 == [str(s) for s in g["strings_utf8"]]


def test_train_two_steps_match_reference():
    g = gu.load("train2")
    chars = gu.alphabet_chars("english")
    hp = dict(input_line_height=30, rds_line_height=30, lstm_input_dim=32, num_lstm_layers=2,
              num_lstm_hidden_units=48, p_lstm_dropout=0.5, num_in_channels=1)
    V = len(chars)
    sd0 = cf.closed_form_state(hp, V)
    sd = vo.state_from_numpy(sd0)
    opt = torch.optim.Adam([p for _, p in vo.trainable(sd)], lr=float(g["lr"]), weight_decay=0.0)
    B = len(g["widths"])
    x, w, tgt, tl = cf.closed_form_batch(B, 1, 30, [int(v) for v in g["widths"]], V,
                                         [int(v) for v in g["labels_per_line"]], seed=1)
    losses = []
    for s in range(int(g["steps"])):
        s1, s2 = cf.closed_form_pool_samples(B, seed=5 + s)
        l, _, _ = vo.train_step(sd, hp, opt, torch.from_numpy(x), w, torch.from_numpy(tgt), torch.from_numpy(tl),
                                (torch.from_numpy(s1), torch.from_numpy(s2)), lstm_training=False)
        losses.append(l)
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-5)
    for k in sd0:
        np.testing.assert_allclose(sd[k].detach().reshape(-1)[:32].numpy(), g["post_head/" + k], rtol=1e-3, atol=2e-5)
        d = (sd[k].detach().double() - torch.from_numpy(sd0[k]).double()).norm().item()
        np.testing.assert_allclose(d, float(g["delta_norm/" + k]), rtol=2e-2, atol=1e-6)
