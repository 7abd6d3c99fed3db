"""CPU: the host-side shell (vistaocr_amd/loop.py) against traces captured from the reference's own
lr_scheduler.py / datautils.py (tests/golden/host_logic.npz), plus the fit() control flow with a stub train step."""
import os

import numpy as np
import torch

from tests import golden_util as gu


def test_plateau_scheduler_matches_reference_traces():
    import vistaocr_amd as va
    from vistaocr_amd.loop import ReduceLROnPlateau
    g = gu.load("host_logic")
    metrics = [float(m) for m in g["sched_metrics"]]
    for name, kw in [("a", dict(patience=2, min_lr=1e-5)), ("b", dict(patience=0, min_lr=1e-4, cooldown=2)), ("c", dict(patience=1, min_lr=0))]:
        p = torch.nn.Parameter(torch.zeros(4))
        opt = va.FlatClampAdam([p], lr=1e-3)              # the build's optimiser exposes param_groups like torch's
        sch = ReduceLROnPlateau(opt, mode="min", **kw)
        trace = []
        for m in metrics:
            r = sch.step(m)
            trace.append((float(bool(r)), float(opt.param_groups[0]["lr"]), float(bool(sch.finished)), float(sch.wait)))
        np.testing.assert_allclose(np.array(trace), g["sched_" + name], rtol=1e-12, atol=0, err_msg=name)


def test_collater_matches_reference_batch():
    from vistaocr_amd.loop import SortByWidthCollater
    g = gu.load("host_logic")
    r = np.random.RandomState(0)
    batch = []
    for i, (w, L) in enumerate([(40, 3), (97, 7), (15, 1), (97, 2), (60, 5)]):
        img = r.uniform(0, 1, size=(1, 30, w + (3 if i == 0 else 0))).astype(np.float32)
        tr = [int(v) for v in r.randint(1, 96, size=L)]
        batch.append((torch.from_numpy(img), tr, {"width": w, "utt-id": "utt%d" % i, "writer-id": 10 + i}))
    x, tgt, widths, tls, meta = SortByWidthCollater(batch)
    assert x.dtype == torch.float32 and tgt.dtype == torch.int32 and widths.dtype == torch.int32 and tls.dtype == torch.int32
    np.testing.assert_array_equal(x.numpy(), g["coll_x"])
    np.testing.assert_array_equal(tgt.numpy(), g["coll_targets"])
    np.testing.assert_array_equal(widths.numpy(), g["coll_widths"])
    np.testing.assert_array_equal(tls.numpy(), g["coll_target_lens"])
    assert meta["utt-ids"] == [str(s) for s in g["coll_ids"]]
    np.testing.assert_array_equal(meta["writer-ids"].numpy(), g["coll_writers"])
    assert widths.tolist() == sorted(widths.tolist(), reverse=True)


def test_grouped_sampler_visits_groups_in_order():
    from vistaocr_amd.loop import GroupedSampler

    class DS:
        size_group_keys = [150, 300, 600]
        size_groups = {150: [4, 7, 9], 300: [], 600: [1, 2, 3, 5]}

        def __len__(self):
            return 7
    torch.manual_seed(0)
    order = list(GroupedSampler(DS()))
    assert sorted(order[:3]) == [4, 7, 9] and sorted(order[3:]) == [1, 2, 3, 5] and len(GroupedSampler(DS())) == 7
    assert list(GroupedSampler(DS(), rand=False)) == [4, 7, 9, 1, 2, 3, 5]
    assert list(GroupedSampler(DS(), rand=False, max_items=4)) == [4, 7, 9, 1]          # ends cleanly (PEP 479)
    s = GroupedSampler(DS(), fixed_rand=True)
    assert list(s) == list(s)


def test_fit_control_flow_and_checkpoint_schema(tmp_path):
    """Validation cadence, plateau -> LR drop -> reload best, early exit, and the checkpoint dictionary."""
    import vistaocr_amd as va
    from vistaocr_amd.loop import fit
    m = va.CnnOcrModel(alphabet=va.english_alphabet(), gpu=False, verbose=False, input_line_height=30, rds_line_height=30,
                       lstm_input_dim=16, num_lstm_layers=1, num_lstm_hidden_units=16, p_lstm_dropout=0.0)
    opt = va.FlatClampAdam(m.parameters(), lr=1e-3)
    wers = iter([0.9, 0.8, 0.85, 0.86, 0.87, 0.88, 0.89, 0.9, 0.91, 0.92])
    marks = []

    def fake_train(batch, model, crit, optimizer):
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)                                  # so that "reload best" is observable
        return 64.0

    def fake_val(loader, model, crit):
        w = next(wers)
        marks.append((w, float(next(model.parameters()).detach().flatten()[0])))
        return 1.0, w / 2, w
    prefix = os.path.join(tmp_path, "run")
    w0 = float(next(m.parameters()).detach().flatten()[0])
    hist = fit(m, None, opt, [None] * 40, None, fake_train, prefix, batch_size=32, n_epochs=1, snapshot_every_n_iterations=4,
               patience=1, min_lr=1e-4, validate_fn=fake_val)
    assert [round(l, 3) for l in hist["loss"][:2]] == [2.0, 2.0]
    assert [v[0] for v in hist["val"]] == [4, 8, 12, 16, 20, 24]          # validation every 4 iterations
    assert hist["lr_drops"] == [16, 24] and hist["stopped_early"]          # patience 1: drop at 16, exhausted at 24
    assert abs(opt.param_groups[0]["lr"] - 1e-4) < 1e-12
    from vistaocr_amd import checkpoint
    ck = checkpoint.load(prefix + "-best_model.pth")
    assert set(ck) == {"iteration", "state_dict", "optimizer", "model_hyper_params", "rtl", "cur_lr", "val_loss", "val_cer", "val_wer", "line_height"}
    assert ck["iteration"] == 8 and abs(ck["val_wer"] - 0.8) < 1e-12       # best = second validation
    # after the LR drop at iteration 16 the best weights (iteration 8 -> w0 + 8) were reloaded before training went on
    assert abs(marks[4][1] - (w0 + 8 + 4)) < 1e-5
    m2 = va.CnnOcrModel.FromSavedWeights(prefix + "-cur_snapshot.pth", verbose=False, gpu=False)
    assert m2.get_hyper_params()["lstm_input_dim"] == 16
