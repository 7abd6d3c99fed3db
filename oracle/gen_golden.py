"""TEST INFRASTRUCTURE ONLY — golden-vector generator.  Runs ONLY in the build container, where the
reference is mounted read-only at /root/reference; it imports the reference's own
`models/cnnlstm.py` (PyTorch-CPU path, with a stub `textutils` because the real one needs icu_bidi
and private files) and writes small .npz fixtures (inputs by closed form, expected
outputs as data) to tests/golden/.  Nothing here travels as source of the reference: fixtures hold
numbers only.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden

Cases: c1 (8x30x300, V=96, train-mode BN), varwidth [300,250,200,15] eval,
rds (60->30), arabic (V=166), h256 (C1 shape with the 256-hidden BiLSTM), train2 (two full
train() steps incl. clamp+Adam), alphabets (symbol tables as data), decode edge cases.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

import numpy as np
import torch

from oracle import closed_form as cf
from oracle import vista_oracle as vo


def import_reference():
    stub = types.ModuleType("textutils")
    stub.uxxxx_to_utf8 = vo.uxxxx_to_utf8          # restates src/textutils.py:216-243
    sys.modules["textutils"] = stub
    sys.path.insert(0, REF)
    from models.cnnlstm import CnnOcrModel           # noqa
    from alphabet import Alphabet                    # noqa
    from english import EnglishAlphabet              # noqa
    from arabic import ArabicAlphabet                # noqa
    from french import FrenchAlphabet                # noqa
    from decoder import ArgmaxDecoder                # noqa
    return CnnOcrModel, Alphabet, EnglishAlphabet, ArabicAlphabet, FrenchAlphabet, ArgmaxDecoder


def build_ref_model(CnnOcrModel, hp, alphabet, sd_np):
    model = CnnOcrModel(alphabet=alphabet, gpu=False, multigpu=False, verbose=False, **hp)
    sd = model.state_dict()
    for k in sd:
        if k not in sd_np:
            continue
        assert tuple(sd[k].shape) == tuple(sd_np[k].shape), (k, sd[k].shape, sd_np[k].shape)
        sd[k] = torch.from_numpy(sd_np[k].copy())
    missing = set(sd_np) - set(sd)
    assert not missing, missing
    model.load_state_dict(sd, strict=True)
    return model


def inject_samples(model, s1, s2):
    pools = [m for m in model.cnn if isinstance(m, torch.nn.FractionalMaxPool2d)]
    assert len(pools) == 2
    pools[0]._random_samples = torch.from_numpy(s1)
    pools[1]._random_samples = torch.from_numpy(s2)


def ref_ctc(logits, targets, act_lens, tgt_lens):
    # north_star's "reference PyTorch-CPU path": warp-ctc is absent, F.ctc_loss has the same definition
    lp = torch.nn.functional.log_softmax(logits, dim=2)
    return torch.nn.functional.ctc_loss(lp, targets.long(), act_lens.long(), tgt_lens.long(), blank=0,
                                        reduction="sum", zero_infinity=False).reshape(1)


def labels_from_strings(model, logits, lens):
    """Integer label sequence = emitted argmax indices; recovered by re-running the reference decode
    in uxxxx mode and mapping through the same per-frame argmax (done by the oracle restatement and
    cross-checked against the reference strings)."""
    strs_ux = model.decode_without_lm(logits, lens, uxxxx=True)
    strs_u8 = model.decode_without_lm(logits, lens, uxxxx=False)
    o_ux, labels = vo.greedy_decode(logits, lens, model.alphabet.idx_to_char, uxxxx=True)
    assert o_ux == strs_ux, "oracle decode restatement disagrees with the reference"
    return strs_ux, strs_u8, labels


def top2_margin(logits, lens):
    T, B, V = logits.shape
    s, _ = torch.sort(logits.detach(), dim=2, descending=True)
    m = (s[:, :, 0] - s[:, :, 1])
    mask = torch.arange(T).unsqueeze(1) < lens.unsqueeze(0)
    return float(m[mask].min())


def grad_summary(model):
    out = {}
    for k, p in model.named_parameters():
        g = p.grad.detach().double()
        out["gnorm/" + k] = np.float64(g.norm().item())
        out["gsum/" + k] = np.float64(g.sum().item())
        out["ghead/" + k] = p.grad.detach().reshape(-1)[:32].numpy().copy()
    return out


def run_case(name, CnnOcrModel, alphabet, hp, B, widths, labels_per_line, mode, himg=None, save_logits=True,
             save_dlogits=False):
    V = len(alphabet)
    sd_np = cf.closed_form_state(hp, V)
    model = build_ref_model(CnnOcrModel, hp, alphabet, sd_np)
    C = hp.get("num_in_channels", 1)
    s1, s2 = cf.closed_form_pool_samples(B)
    inject_samples(model, s1, s2)
    if mode == "train":
        model.train()
        model.lstm.eval()          # BN batch stats, LSTM dropout off (SURVEY.md §7 'LSTM dropout parity')
    else:
        # eval mode needs running statistics that fit the data: converge them with 25 train-mode BN
        # passes over a fixed warm-up batch, and ship them in the fixture (a few KB) as state overrides
        xw, ww, _, _ = cf.closed_form_batch(B, C, hp["input_line_height"], widths, V, labels_per_line, seed=99)
        model.train()
        model.lstm.eval()
        with torch.no_grad():
            for _ in range(25):
                model(torch.from_numpy(xw), torch.from_numpy(ww))
        model.eval()
    # pick the first batch seed whose greedy top1-top2 margin is comfortably above fp32 reordering
    # noise, so that "bit-exact labels" is a meaningful, stable check (SURVEY.md §7 hard parts)
    best = None
    saved_rs = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k}
    for seed in range(1, 13):
        x, w, tgt, tl = cf.closed_form_batch(B, C, hp["input_line_height"], widths, V, labels_per_line, seed=seed)
        with torch.no_grad():
            lg, ln = model(torch.from_numpy(x), torch.from_numpy(w))
        mg = top2_margin(lg, ln)
        if best is None or mg > best[0]:
            best = (mg, seed)
        if mg >= 2e-3:
            break
    model.load_state_dict({**model.state_dict(), **saved_rs})      # undo running-stat updates of the search
    seed = best[1]
    x, w, tgt, tl = cf.closed_form_batch(B, C, hp["input_line_height"], widths, V, labels_per_line, seed=seed)
    taps = {}
    hooks = []
    for idx, m in enumerate(model.cnn):
        if isinstance(m, (torch.nn.ReLU, torch.nn.FractionalMaxPool2d)):
            hooks.append(m.register_forward_hook(lambda mod, i, o, idx=idx: taps.__setitem__("cnn%d" % idx, o.detach().clone())))
    hooks.append(model.bridge_layer.register_forward_hook(lambda mod, i, o: taps.__setitem__("bridge", o.detach().clone())))
    xt = torch.from_numpy(x)
    logits, lens = model(xt, torch.from_numpy(w))
    logits.retain_grad()
    loss = ref_ctc(logits, torch.from_numpy(tgt), lens, torch.from_numpy(tl))
    out = {
        "hp_json": np.array(repr(sorted(hp.items()))),
        "mode": np.array(mode), "B": np.int64(B), "batch_seed": np.int64(seed), "widths": w, "labels_per_line": tl,
        "lens": lens.numpy().copy(), "loss": loss.detach().numpy().copy(),
        "margin": np.float64(top2_margin(logits, lens)),
        "logits_shape": np.array(logits.shape),
    }
    if mode != "train":
        for k, v in model.state_dict().items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                out["state/" + k] = v.numpy().copy()
    if mode == "train":
        loss.backward()
        out.update(grad_summary(model))
        if save_dlogits:
            out["dlogits"] = logits.grad.numpy().copy()
        out["dlogits_sum_abs"] = np.float64(logits.grad.double().abs().sum().item())
        for k, v in model.state_dict().items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                out["post/" + k] = v.numpy().copy()
    strs_ux, strs_u8, labels = labels_from_strings(model, logits.detach(), lens)
    out["strings_uxxxx"] = np.array(strs_ux, dtype=object)
    out["strings_utf8"] = np.array(strs_u8, dtype=object)
    out["labels_flat"] = np.array([v for l in labels for v in l], dtype=np.int32)
    out["labels_len"] = np.array([len(l) for l in labels], dtype=np.int32)
    if save_logits:
        out["logits"] = logits.detach().numpy().copy()
    else:
        out["logits_head"] = logits.detach()[:, :, :8].numpy().copy()
        out["logits_argmax"] = logits.detach().argmax(2).to(torch.int32).numpy().copy()
        out["logits_max"] = logits.detach().max(2)[0].numpy().copy()
    for k, v in taps.items():
        vd = v.double()
        out["tap_sum/" + k] = np.float64(vd.sum().item())
        out["tap_abs/" + k] = np.float64(vd.abs().sum().item())
        out["tap_shape/" + k] = np.array(v.shape)
        out["tap_crop/" + k] = v.reshape(-1)[:: max(1, v.numel() // 257)][:257].numpy().copy()
    for h in hooks:
        h.remove()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("%-10s loss=%.6f lens=%s margin=%.4g labels/line=%s" % (name, float(loss), lens.tolist()[:4], out["margin"],
                                                                   out["labels_len"].tolist()[:4]))


def run_train2(CnnOcrModel, alphabet, hp, B, widths, labels_per_line, steps=2, lr=1e-3):
    """Two reference train() steps (train_cnn_lstm.py:131-150) with LSTM dropout off."""
    V = len(alphabet)
    sd_np = cf.closed_form_state(hp, V)
    model = build_ref_model(CnnOcrModel, hp, alphabet, sd_np)
    model.train()
    model.lstm.eval()
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=0.0)
    x, w, tgt, tl = cf.closed_form_batch(B, 1, hp["input_line_height"], widths, V, labels_per_line, seed=1)
    out = {"widths": w, "labels_per_line": tl, "lr": np.float64(lr), "steps": np.int64(steps)}
    losses = []
    for s in range(steps):
        s1, s2 = cf.closed_form_pool_samples(B, seed=5 + s)
        inject_samples(model, s1, s2)
        opt.zero_grad()
        logits, lens = model(torch.from_numpy(x), torch.from_numpy(w))
        loss = ref_ctc(logits, torch.from_numpy(tgt), lens, torch.from_numpy(tl))
        loss.backward()
        for p in model.parameters():
            if p.grad is not None:
                p.grad.data.clamp_(min=-5, max=5)
        opt.step()
        losses.append(float(loss))
    out["losses"] = np.array(losses, dtype=np.float64)
    for k, v in model.state_dict().items():
        if k not in sd_np:
            continue
        d = v.double() - torch.from_numpy(sd_np[k]).double()
        out["delta_norm/" + k] = np.float64(d.norm().item())
        out["post_head/" + k] = v.reshape(-1)[:32].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "train2.npz"), **out)
    print("train2     losses=%s" % losses)


def run_decode_edges(CnnOcrModel, ArgmaxDecoder, alphabet):
    """Hand-built logits that exercise every branch of decode_without_lm / ArgmaxDecoder.decode:
    blanks, sub-threshold maxima, repeated labels, the duplicated 'u002d' entries (English idx 73
    and 91), ties (first max wins), frames past `lens`, and a label on the very last frame."""
    V = len(alphabet)
    T, B = 12, 4
    lg = cf.fill((T, B, V), "decode_edges", 0.01).copy()
    def put(t, b, k, v=1.0):
        lg[t, b, k] = v
    for t, k in enumerate([5, 5, 0, 5, 73, 91, 73, 7, 7, 0, 9, 9]):
        put(t, 0, k)
    for t, k in enumerate([3, 4, 4, 4, 0, 0, 4, 2, 2, 2, 1, 0]):
        put(t, 1, k)
    for t in range(T):
        put(t, 2, 10 + t, 0.031 if t % 2 == 0 else 0.032)     # around 3/96 = 0.03125
    lg[0, 3, 20] = lg[0, 3, 40] = 2.0                            # tie: first index wins
    lg[1, 3, 40] = 2.0
    lg[2, 3, 0] = 3.0
    lg[11, 3, 33] = 2.0
    lens = torch.tensor([12, 10, 12, 12], dtype=torch.int32)
    logits = torch.from_numpy(lg)
    hp = dict(input_line_height=30, rds_line_height=30, lstm_input_dim=8, num_lstm_layers=1,
              num_lstm_hidden_units=8, p_lstm_dropout=0.0)
    model = CnnOcrModel(alphabet=alphabet, gpu=False, multigpu=False, verbose=False, **hp)
    a = model.decode_without_lm(logits, lens, uxxxx=True)
    b = ArgmaxDecoder(alphabet).decode(logits, lens, uxxxx=True)
    c = model.decode_without_lm(logits, lens, uxxxx=False)
    assert a == b
    _, labels = vo.greedy_decode(logits, lens, alphabet.idx_to_char, uxxxx=True)
    np.savez_compressed(os.path.join(OUT, "decode_edges.npz"), logits=lg, lens=lens.numpy(),
                        strings_uxxxx=np.array(a, dtype=object), strings_utf8=np.array(c, dtype=object),
                        labels_flat=np.array([v for l in labels for v in l], dtype=np.int32),
                        labels_len=np.array([len(l) for l in labels], dtype=np.int32))
    print("decode_edges", a)


def run_host_logic():
    """Host-side control logic of the reference that the build mirrors: the plateau scheduler
    (src/lr_scheduler.py) driven by metric sequences, and SortByWidthCollater (src/datautils.py) on a small batch."""
    from lr_scheduler import ReduceLROnPlateau
    from datautils import SortByWidthCollater
    out = {}
    metrics = [5.0, 4.0, 4.0, 4.00005, 3.9, 3.95, 3.95, 3.95, 3.95, 3.0, 3.1, 3.1, 3.1, 3.1, 3.1, 3.1, 3.1, 3.1, 3.1, 3.1]
    for name, kw in [("a", dict(patience=2, min_lr=1e-5)), ("b", dict(patience=0, min_lr=1e-4, cooldown=2)), ("c", dict(patience=1, min_lr=0))]:
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=1e-3)
        sch = ReduceLROnPlateau(opt, mode="min", **kw)
        trace = []
        for m in metrics:
            r = sch.step(m)
            trace.append((float(bool(r)), float(opt.param_groups[0]["lr"]), float(bool(sch.finished)), float(sch.wait)))
        out["sched_" + name] = np.array(trace, dtype=np.float64)
    out["sched_metrics"] = np.array(metrics, dtype=np.float64)
    r = np.random.RandomState(0)
    batch, spec = [], []
    for i, (w, L) in enumerate([(40, 3), (97, 7), (15, 1), (97, 2), (60, 5)]):
        img = r.uniform(0, 1, size=(1, 30, w + (3 if i == 0 else 0))).astype(np.float32)     # item 0 arrives pre-padded
        tr = [int(v) for v in r.randint(1, 96, size=L)]
        batch.append((torch.from_numpy(img), tr, {"width": w, "utt-id": "utt%d" % i, "writer-id": 10 + i}))
        spec.append((w, L))
    x, tgt, widths, tls, meta = SortByWidthCollater(list(batch))
    out["coll_x"] = x.numpy()
    out["coll_targets"] = tgt.numpy()
    out["coll_widths"] = widths.numpy()
    out["coll_target_lens"] = tls.numpy()
    out["coll_ids"] = np.array(meta["utt-ids"], dtype=object)
    out["coll_writers"] = meta["writer-ids"].numpy()
    np.savez_compressed(os.path.join(OUT, "host_logic.npz"), **out)
    print("host_logic  scheduler traces + collater batch saved")


def run_ref_checkpoint(CnnOcrModel, alphabet):
    """A checkpoint exactly as the reference writes it (src/train_cnn_lstm.py:427-438) from a model trained with the
    reference's defaults gpu=True, multigpu=True: the hyper-parameter dict pickles the reference's own
    `alphabet.Alphabet` instance, the CNN keys carry nn.DataParallel's `cnn.module.` prefix, the optimiser state is
    torch.optim.Adam's.  Stored gzip-compressed (tests/golden/ref_checkpoint.pth.gz); to keep it small the five large
    conv weights keep one entry in sixteen (the file pins FORMAT interop; numerics are pinned by the other goldens).
    Expected outputs: the reference model's eval-mode logits / greedy strings on a fixed closed-form batch."""
    import gzip
    import io
    hp = dict(input_line_height=30, rds_line_height=30, lstm_input_dim=16, num_lstm_layers=1,
              num_lstm_hidden_units=16, p_lstm_dropout=0.5, num_in_channels=1)
    V = len(alphabet)
    sd_np = cf.closed_form_state(hp, V)
    for k in ("cnn.7.weight", "cnn.10.weight", "cnn.14.weight", "cnn.17.weight", "cnn.20.weight"):
        flat = sd_np[k].reshape(-1)
        keep = (np.arange(flat.size) % 16) == 3
        sd_np[k] = (flat * keep * np.float32(4.0)).reshape(sd_np[k].shape).astype(np.float32)
    model = build_ref_model(CnnOcrModel, hp, alphabet, sd_np)
    # what a gpu=True, multigpu=True run holds: DataParallel around the CNN (cnnlstm.py:198-199) and those kwargs in the
    # hyper-parameter dict (cnnlstm.py:80: self.hyper_params = kwargs.copy())
    model.cnn = torch.nn.DataParallel(model.cnn)
    model.hyper_params = dict(hp, alphabet=alphabet, gpu=True, multigpu=True, verbose=False)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=0.0)
    B, widths = 2, [120, 90]
    x, w, tgt, tl = cf.closed_form_batch(B, 1, 30, widths, V, [4, 3], seed=2)
    s1, s2 = cf.closed_form_pool_samples(B)
    pools = [m for m in model.cnn.module if isinstance(m, torch.nn.FractionalMaxPool2d)]
    pools[0]._random_samples = torch.from_numpy(s1)
    pools[1]._random_samples = torch.from_numpy(s2)
    model.eval()
    with torch.no_grad():
        logits, lens = model(torch.from_numpy(x), torch.from_numpy(w))
    strs_ux = model.decode_without_lm(logits, lens, uxxxx=True)
    for pl in pools:                   # the injected samples are test plumbing, not part of a trained model's state
        pl._random_samples = None
    ckpt = {'iteration': 1234, 'state_dict': model.state_dict(), 'optimizer': opt.state_dict(),
            'model_hyper_params': model.get_hyper_params(), 'rtl': False, 'cur_lr': 1e-3, 'val_loss': 1.5,
            'val_cer': 0.25, 'val_wer': 0.5, 'line_height': 30}
    assert any(k.startswith("cnn.module.") for k in ckpt['state_dict'])
    buf = io.BytesIO()
    torch.save(ckpt, buf)
    with gzip.GzipFile(os.path.join(OUT, "ref_checkpoint.pth.gz"), "wb", mtime=0) as fh:
        fh.write(buf.getvalue())
    np.savez_compressed(os.path.join(OUT, "ref_checkpoint_expect.npz"), widths=w, lens=lens.numpy(), logits=logits.numpy(),
                        strings_uxxxx=np.array(strs_ux, dtype=object), batch_seed=np.int64(2),
                        margin=np.float64(top2_margin(logits, lens)), n_keys=np.int64(len(ckpt['state_dict'])))
    print("ref_checkpoint  %d keys, %.0f KB gz, margin %.3g, strings %s" %
          (len(ckpt['state_dict']), os.path.getsize(os.path.join(OUT, "ref_checkpoint.pth.gz")) / 1024.0,
           top2_margin(logits, lens), strs_ux))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    torch.manual_seed(0)
    CnnOcrModel, Alphabet, EnglishAlphabet, ArabicAlphabet, FrenchAlphabet, ArgmaxDecoder = import_reference()
    eng = Alphabet(EnglishAlphabet(None).idx_to_alphabet, left_to_right=True)
    ara = Alphabet(ArabicAlphabet(None).idx_to_alphabet, left_to_right=False)
    import json, tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as fh:
        json.dump({"train": [], "validation": [], "test": []}, fh)     # french.py:14-19 wants a desc.json
    fre = Alphabet(FrenchAlphabet(None, fh.name).idx_to_alphabet, left_to_right=True)
    os.unlink(fh.name)
    if len(sys.argv) > 1 and sys.argv[1] == "ref_checkpoint":          # regenerate this one fixture only
        run_ref_checkpoint(CnnOcrModel, eng)
        return
    np.savez_compressed(os.path.join(OUT, "alphabets.npz"),
                        english=np.array(eng.char_array, dtype=object),
                        arabic=np.array(ara.char_array, dtype=object),
                        french=np.array(fre.char_array, dtype=object))
    # width -> T table through the reference's own size calculator (cnnlstm.py:250-260)
    hp30 = dict(input_line_height=30, rds_line_height=30, lstm_input_dim=16, num_lstm_layers=1,
                num_lstm_hidden_units=8, p_lstm_dropout=0.0)
    m30 = CnnOcrModel(alphabet=eng, gpu=False, multigpu=False, verbose=False, **hp30)
    ws = np.array(list(range(15, 1301)), dtype=np.int64)
    t30 = np.array([m30.cnn_input_size_to_output_size((30, int(w)))[1] for w in ws], dtype=np.int64)
    hp60 = dict(hp30, input_line_height=60)
    m60 = CnnOcrModel(alphabet=eng, gpu=False, multigpu=False, verbose=False, **hp60)
    t60 = np.array([m60.cnn_input_size_to_output_size((60, int(w)))[1] for w in ws], dtype=np.int64)
    h30 = m30.cnn_input_size_to_output_size((30, 20))[0]
    h60 = m60.cnn_input_size_to_output_size((60, 20))[0]
    np.savez_compressed(os.path.join(OUT, "width_table.npz"), widths=ws, t30=t30, t60=t60,
                        h30=np.int64(h30), h60=np.int64(h60))

    small = dict(input_line_height=30, rds_line_height=30, lstm_input_dim=32, num_lstm_layers=2,
                 num_lstm_hidden_units=48, p_lstm_dropout=0.5, num_in_channels=1)
    run_case("c1", CnnOcrModel, eng, small, 8, [300] * 8, [12] * 8, "train", save_dlogits=True)
    run_case("c1_eval", CnnOcrModel, eng, small, 8, [300] * 8, [12] * 8, "eval")
    run_case("varwidth", CnnOcrModel, eng, small, 4, [300, 250, 200, 15], [10, 8, 6, 1], "eval")
    run_case("varwidth_train", CnnOcrModel, eng, small, 4, [300, 250, 200, 15], [10, 8, 6, 1], "train")
    rds = dict(small, input_line_height=60)
    run_case("rds", CnnOcrModel, eng, rds, 2, [240, 200], [6, 5], "train")
    run_case("arabic", CnnOcrModel, ara, small, 2, [260, 180], [9, 7], "train")
    h256 = dict(small, lstm_input_dim=128, num_lstm_layers=3, num_lstm_hidden_units=256)
    run_case("h256", CnnOcrModel, eng, h256, 8, [300] * 8, [12] * 8, "train", save_logits=False)
    run_train2(CnnOcrModel, eng, small, 4, [200, 200, 160, 120], [8, 8, 6, 4])
    run_decode_edges(CnnOcrModel, ArgmaxDecoder, eng)
    run_host_logic()
    run_ref_checkpoint(CnnOcrModel, eng)
    # no bytecode may be left in the read-only reference tree
    for d in (REF, os.path.join(REF, "models")):
        assert not os.path.exists(os.path.join(d, "__pycache__")), d


if __name__ == "__main__":
    main()
