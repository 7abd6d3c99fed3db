"""CPU: host-side logic of the drop-in surface — alphabets, size bookkeeping, state-dict keys, ctor errors,
text utilities (incl. the reference's one known-answer test), checkpoint schema."""
import os

import numpy as np
import pytest
import torch

from oracle import closed_form as cf
from tests import golden_util as gu

HP = dict(input_line_height=30, rds_line_height=30, lstm_input_dim=32, num_lstm_layers=2, num_lstm_hidden_units=48,
          p_lstm_dropout=0.5)


def test_alphabets_match_reference_tables():
    import vistaocr_amd as va
    for name, fn, ltr in [("english", va.english_alphabet, True), ("arabic", va.arabic_alphabet, False), ("french", va.french_alphabet, True)]:
        a = fn()
        assert a.char_array == gu.alphabet_chars(name)
        assert a.left_to_right == ltr and a.idx_to_char[0] == "<ctc-blank>"
    e = va.english_alphabet()
    assert len(e) == 96 and len(set(e.char_array)) == 95
    assert e.idx_to_char[73] == e.idx_to_char[91] == "u002d" and e.char_to_idx["u002d"] == 91   # last duplicate wins
    assert e.canonical_indices()[91] == 73
    assert len(va.arabic_alphabet()) == 166 and len(va.french_alphabet()) == 101


def test_state_dict_keys_and_shapes_match_reference():
    import vistaocr_amd as va
    for hp, V in [(HP, 96), (dict(HP, input_line_height=60), 96), (dict(HP, num_lstm_layers=3, num_lstm_hidden_units=256), 166)]:
        al = va.Alphabet(["<ctc-blank>"] + ["u%04x" % (0x61 + i) for i in range(V - 1)])
        m = va.CnnOcrModel(alphabet=al, gpu=False, verbose=False, **hp)
        sd = m.state_dict()
        want = dict(cf.param_shapes(hp, V))
        got = {k: tuple(v.shape) for k, v in sd.items() if not k.endswith("num_batches_tracked")}
        assert got == want
        assert sum(1 for k in sd if k.endswith("num_batches_tracked")) == 7
        vals = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
        assert float(vals.min()) >= -0.08 and float(vals.max()) <= 0.08          # cnnlstm.py:158-159
    big = va.CnnOcrModel(alphabet=va.english_alphabet(), gpu=False, verbose=False, input_line_height=30, rds_line_height=30,
                         lstm_input_dim=128, num_lstm_layers=3, num_lstm_hidden_units=512, p_lstm_dropout=0.5)
    assert sum(p.numel() for p in big.parameters()) == 17293472                  # SURVEY.md §8 a-1 [probe]
    assert len(big.state_dict()) == 77


def test_ctor_errors_and_hyper_params():
    import vistaocr_amd as va
    al = va.english_alphabet()
    with pytest.raises(Exception, match="Only keyword arguments"):
        va.CnnOcrModel(30)
    with pytest.raises(Exception, match="less than or equal"):
        va.CnnOcrModel(alphabet=al, gpu=False, verbose=False, **dict(HP, rds_line_height=60))
    with pytest.raises(Exception, match="power of 2"):
        va.CnnOcrModel(alphabet=al, gpu=False, verbose=False, **dict(HP, input_line_height=90))
    with pytest.raises(Exception, match="power of 2"):
        va.CnnOcrModel(alphabet=al, gpu=False, verbose=False, **dict(HP, input_line_height=45, rds_line_height=15))
    m = va.CnnOcrModel(alphabet=al, gpu=False, verbose=False, **HP)
    hp = m.get_hyper_params()
    assert hp["alphabet"] is al and hp["lstm_input_dim"] == 32 and hp["gpu"] is False


def test_width_table_matches_reference():
    import vistaocr_amd as va
    t = gu.load("width_table")
    al = va.english_alphabet()
    m30 = va.CnnOcrModel(alphabet=al, gpu=False, verbose=False, **HP)
    m60 = va.CnnOcrModel(alphabet=al, gpu=False, verbose=False, **dict(HP, input_line_height=60))
    for w, a, b in zip(t["widths"], t["t30"], t["t60"]):
        assert m30.cnn_input_size_to_output_size((30, int(w)))[1] == int(a)
        assert m60.cnn_input_size_to_output_size((60, int(w)))[1] == int(b)
    assert m30.cnn_input_size_to_output_size((30, 20))[0] == int(t["h30"])
    assert m60.cnn_input_size_to_output_size((60, 20))[0] == int(t["h60"])


def test_checkpoint_roundtrip_schema(tmp_path):
    """FromSavedWeights reads the reference's checkpoint dict (train_cnn_lstm.py:427-438), incl. the DataParallel
    'cnn.module.' key prefix of multigpu checkpoints (utils/decode.py:58-71)."""
    import vistaocr_amd as va
    al = va.english_alphabet()
    m = va.CnnOcrModel(alphabet=al, gpu=False, verbose=False, **HP)
    sd = {(k.replace("cnn.", "cnn.module.", 1) if k.startswith("cnn.") else k): v for k, v in m.state_dict().items()}
    ck = dict(iteration=7, state_dict=sd, optimizer={}, model_hyper_params=m.get_hyper_params(), rtl=False, cur_lr=1e-3,
              val_loss=1.0, val_cer=0.5, val_wer=0.9, line_height=30)
    path = os.path.join(tmp_path, "x-cur_snapshot.pth")
    torch.save(ck, path)
    m2 = va.CnnOcrModel.FromSavedWeights(path, verbose=False, gpu=False)
    assert m2.rtl is False
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k])


def _ref_checkpoint(tmp_path):
    import gzip
    path = os.path.join(tmp_path, "ref-best_model.pth")
    with gzip.open(os.path.join(gu.GOLDEN, "ref_checkpoint.pth.gz"), "rb") as src, open(path, "wb") as dst:
        dst.write(src.read())
    return path


def test_reference_written_checkpoint_loads(tmp_path):
    """tests/golden/ref_checkpoint.pth.gz was written by the REFERENCE's own classes (oracle/gen_golden.py
    run_ref_checkpoint: train_cnn_lstm.py:427-438 schema, pickled `alphabet.Alphabet`, DataParallel 'cnn.module.' keys,
    torch.optim.Adam state).  A plain torch.load cannot even unpickle it here (no top-level module `alphabet`);
    FromSavedWeights must — without leaving anything in sys.modules."""
    import sys
    import vistaocr_amd as va
    from vistaocr_amd import checkpoint
    path = _ref_checkpoint(tmp_path)
    with pytest.raises(ModuleNotFoundError):
        torch.load(path, map_location="cpu", weights_only=False)
    w = checkpoint.load(path)
    assert "alphabet" not in sys.modules
    assert set(w) == {"iteration", "state_dict", "optimizer", "model_hyper_params", "rtl", "cur_lr", "val_loss", "val_cer",
                      "val_wer", "line_height"}
    al = w["model_hyper_params"]["alphabet"]
    assert isinstance(al, va.Alphabet) and len(al) == 96 and al.left_to_right is True
    assert al.idx_to_char == va.english_alphabet().idx_to_char and al.char_to_idx == va.english_alphabet().char_to_idx
    assert sum(k.startswith("cnn.module.") for k in w["state_dict"]) == 49          # 7 x (conv w,b + BN 5 entries)
    m = va.CnnOcrModel.FromSavedWeights(path, verbose=False, gpu=False)
    exp = gu.load("ref_checkpoint_expect")
    assert len(m.state_dict()) == int(exp["n_keys"]) and m.rtl is False and m.num_lstm_hidden_units == 16
    for k, v in checkpoint.strip_dataparallel_prefix(w["state_dict"]).items():
        assert torch.equal(m.state_dict()[k], v), k
    # and back: a snapshot this build writes names the alphabet class the reference's way and re-adds the prefix
    from vistaocr_amd.loop import save_snapshot

    class _Opt:
        def state_dict(self):
            return {}
    m.hyper_params["gpu"] = True
    out = os.path.join(tmp_path, "ours-cur_snapshot.pth")
    save_snapshot(out, 5, m, _Opt(), False, 1e-3, 1.0, 0.5, 0.9, 30)
    import zipfile
    z = zipfile.ZipFile(out)
    pkl = z.read([n for n in z.namelist() if n.endswith("data.pkl")][0])
    assert b"alphabet\nAlphabet\n" in pkl and b"vistaocr_amd" not in pkl
    w2 = checkpoint.load(out)
    assert sorted(w2["state_dict"]) == sorted(w["state_dict"])
    m.hyper_params["gpu"] = False
    m2 = va.CnnOcrModel.FromSavedWeights(out, verbose=False, gpu=False)
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k])


def test_textutils_and_edit_distance_kat():
    from vistaocr_amd import textutils as tu
    assert tu.uxxxx_to_utf8("u0061 u0062 u0020 u0063") == "ab c"
    assert tu.uxxxx_to_utf8("   ") == "" and tu.uxxxx_to_utf8("<unk> u0041") == "<unk>A"
    assert tu.utf8_to_uxxxx("aB ") == "u0061 u0042 u0020"
    assert tu.utf8_to_uxxxx("ا", output_array=True) == ["u0627"]
    # the reference's only known-answer test: src/edit_dist_trace.py:72-83 (dist = 13)
    ref = "\" McNamara's Band , \" \" Greensleeves \" and \" English Rose . \""
    hyp = " ' He Namarod's Layd , \" \" breensleeres \" and \" English hose . '"
    assert tu.edit_distance(hyp, ref) == 13
    assert tu.edit_distance("", "") == 0 and tu.edit_distance("abc", "") == 3 and tu.edit_distance([], ["a"]) == 1
    r = np.random.RandomState(0)
    for _ in range(50):
        a = list(r.randint(0, 4, size=r.randint(0, 12)))
        b = list(r.randint(0, 4, size=r.randint(0, 12)))
        d = np.zeros((len(a) + 1, len(b) + 1))
        d[:, 0] = np.arange(len(a) + 1)
        d[0, :] = np.arange(len(b) + 1)
        for i in range(1, len(a) + 1):
            for j in range(1, len(b) + 1):
                d[i, j] = d[i - 1, j - 1] if a[i - 1] == b[j - 1] else 1 + min(d[i, j - 1], d[i - 1, j], d[i - 1, j - 1])
        assert tu.edit_distance(a, b) == d[-1, -1]
    words = tu.form_tokenized_words("u0061 u0062 u0020 u0063 u002e u0031 u0064".split())
    assert words == ["u0061_u0062", "u0063", "u002e", "u0031", "u0064"]
    cer, wer = tu.compute_cer_wer("u0061 u0062 u0020 u0063", "u0061 u0062 u0020 u0064")
    assert cer == 0.25 and wer == 0.5
    import vistaocr_amd as va
    assert tu.form_target_transcription([1, 2, 63], va.english_alphabet()) == "u0061 u0062 u0020"


def test_optimizer_state_moves_between_flat_adam_and_torch_adam():
    """The 'optimizer' entry of a snapshot (src/train_cnn_lstm.py:429 stores torch.optim.Adam's state_dict): FlatClampAdam writes and
    reads that format, so its moments load into a torch.optim.Adam over the same model and the other way round."""
    import vistaocr_amd as va
    kw = dict(alphabet=va.english_alphabet(), gpu=False, verbose=False, input_line_height=30, rds_line_height=30, lstm_input_dim=16,
              num_lstm_layers=1, num_lstm_hidden_units=16, p_lstm_dropout=0.0)
    torch.manual_seed(0)
    ma, mb = va.CnnOcrModel(**kw), va.CnnOcrModel(**kw)
    flat = va.FlatClampAdam(ma.parameters(), lr=3e-4, weight_decay=1e-5)
    flat.step_count = 7
    flat.exp_avg.copy_(torch.arange(flat.exp_avg.numel(), dtype=torch.float32) * 1e-3)
    flat.exp_avg_sq.copy_(torch.arange(flat.exp_avg_sq.numel(), dtype=torch.float32) * 1e-6 + 1.0)
    sd = flat.state_dict()
    assert set(sd) == {"state", "param_groups"} and sd["param_groups"][0]["params"] == list(range(len(list(ma.parameters()))))
    adam = torch.optim.Adam(mb.parameters(), lr=1e-3)
    adam.load_state_dict(sd)                                            # torch's own loader accepts it
    assert adam.param_groups[0]["lr"] == 3e-4 and adam.param_groups[0]["weight_decay"] == 1e-5
    off = 0
    for p in mb.parameters():
        st = adam.state[p]
        assert float(st["step"]) == 7.0
        assert torch.equal(st["exp_avg"].reshape(-1), flat.exp_avg[off:off + p.numel()])
        assert torch.equal(st["exp_avg_sq"].reshape(-1), flat.exp_avg_sq[off:off + p.numel()])
        off += p.numel()
    # and back: a torch.optim.Adam state into a fresh FlatClampAdam
    mc = va.CnnOcrModel(**kw)
    flat2 = va.FlatClampAdam(mc.parameters(), lr=1e-3)
    flat2.load_state_dict(adam.state_dict())
    assert flat2.step_count == 7 and flat2.param_groups[0]["lr"] == 3e-4
    assert torch.equal(flat2.exp_avg, flat.exp_avg) and torch.equal(flat2.exp_avg_sq, flat.exp_avg_sq)
    # round-1 snapshots (flat moments) still load
    flat2.load_state_dict(dict(step=3, exp_avg=flat.exp_avg * 2, exp_avg_sq=flat.exp_avg_sq, param_groups=[dict(lr=5e-4)]))
    assert flat2.step_count == 3 and flat2.param_groups[0]["lr"] == 5e-4


def test_bench_flop_accounting_of_the_minimal_filtering_kernels():
    """bench.py's roofline leg: algorithmic FLOPs of a conv launch from its call arguments, and the share of them the kernel that the
    library's dispatch picks actually executes (`achieved` is algorithmic, `executed_*` is reported beside it)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    args = lambda n, cin, h, w, cout: (0, 0, 0, 0, n, cin, h, w, cout, 0)
    assert b.conv_flops(args(32, 256, 7, 294, 256)) == 2.0 * 32 * 7 * 294 * 256 * 256 * 9
    # forward / data gradient: F(4,3) along the row from 64 output channels on (6 of 12 multiplications), F(2,3) below (4 of 6)
    assert b.executed_share("vocr_conv3x3_wino_fwd", args(32, 128, 15, 420, 128)) == 0.5
    assert b.executed_share("vocr_conv3x3_wino_fwd", args(32, 64, 30, 600, 64)) == 0.5
    assert b.executed_share("vocr_conv3x3_wino_fwd", args(32, 64, 30, 600, 16)) == pytest.approx(2.0 / 3.0)
    # weight gradient: F(3,2) along the row and across row pairs: 16 of 36, times 2 ceil(H/2) / H for an odd height
    assert b.executed_share("vocr_conv3x3_wgrad_wino", args(32, 64, 30, 600, 64)) == pytest.approx(4.0 / 9.0)
    assert b.executed_share("vocr_conv3x3_wgrad_wino", args(32, 256, 7, 294, 256)) == pytest.approx(4.0 / 9.0 * 8.0 / 7.0)
    assert b.executed_share("vocr_conv3x3_wgrad_wino", args(32, 5, 7, 294, 7)) == pytest.approx(2.0 / 3.0)      # Cin * Cout % 4 != 0: one-row kernel
    assert b.executed_share("vocr_gemm", (0,) * 10) == 1.0
    assert sorted(b.WORKLOADS) == ["c1", "c4", "c5"]


def test_packed_row_count_is_the_layouts_formula():
    """include/vocr.h: rows = 4 * (sum of the chains' longest lengths + chains + 1); a uniform batch of B % 4 == 0 lines packs to more rows
    than T*B (the zero groups), which is why CnnOcrModel keeps such a batch dense (pack_threshold)."""
    from vistaocr_amd import ops
    assert ops.packed_row_count([9, 9, 7, 7, 7, 3, 2, 2, 1], 9) == 4 * (9 + 7 + 1 + 3 + 1)
    assert ops.packed_row_count([294] * 32, 32) == 4 * (8 * 294 + 9) > 294 * 32
    assert ops.packed_row_count([12], 1) == 4 * (12 + 2)
