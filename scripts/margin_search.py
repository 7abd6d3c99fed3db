"""CPU (build container or GPU box host): which closed-form batch seeds give the full-size configuration (B=32, 30x600, 3xBiLSTM-512)
a comfortable greedy top-2 margin on the ORACLE, and how many labels the lines emit.  Used to pick the default seed of
tests/test_round2_gpu.py::test_config1_full_size_vs_oracle and bench.py's parity leg (the tests repeat the search on the box,
starting from that seed, so a different host CPU cannot make the check vacuous)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import closed_form as cf
from oracle import vista_oracle as vo
from tests import golden_util as gu
from tests.parity_util import decode_margin


if len(sys.argv) > 1 and sys.argv[1] == "cases":          # oracle-side seed search of every _run_pair case of tests/test_configs_gpu.py
    from tests import test_configs_gpu as tc
    torch.set_num_threads(os.cpu_count() or 1)
    for name, case in tc.CASES.items():
        if case.get("min_label_agreement", 1.0) < 1.0:
            continue
        t0 = time.time()
        try:
            print("%-16s first seed %d -> seed %d, oracle decode margin %.2e (%.1f s)" % ((name, case["seed"]) + tc.pick_case_seed(case) + (time.time() - t0,)), flush=True)
        except AssertionError as e:
            print("%-16s %s" % (name, e), flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "bench":          # the parity leg of bench.py: its own pool samples, dropout off;  bench <first> <last> [c1|c4|c5]
    import bench
    torch.set_num_threads(os.cpu_count() or 1)
    wl = bench.select_workload(sys.argv[4] if len(sys.argv) > 4 else "c1")
    V = len(gu.alphabet_chars(wl["alphabet"]))
    for seed in range(int(sys.argv[2]), int(sys.argv[3])):
        bench.PARITY_BATCH_SEED = seed
        sd_np, x, w, tgt, tl = bench.parity_inputs(512, V)
        t0 = time.time()
        with torch.no_grad():
            lo, ln = vo.forward(vo.state_from_numpy(sd_np, requires_grad=False), dict(wl["hp"]), torch.from_numpy(x), w, bench.parity_samples(),
                                training=True, lstm_training=False)
        print("bench %s parity seed %d: oracle decode margin %.3e  (%.1f s)" % (sys.argv[4] if len(sys.argv) > 4 else "c1", seed, decode_margin(lo, ln, V), time.time() - t0), flush=True)
    sys.exit(0)
prob_scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
blank_bias = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
seeds = range(int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else range(21, 33)
dropout = (sys.argv[5] == "masks") if len(sys.argv) > 5 else True
chars = gu.alphabet_chars("english")
V = len(chars)
hp = dict(input_line_height=30, rds_line_height=30, lstm_input_dim=128, num_lstm_layers=3, num_lstm_hidden_units=512,
          p_lstm_dropout=0.5, num_in_channels=1)
B, T, H = 32, 294, 512
sd_np = cf.closed_form_state(hp, V, lstm_scale=0.08, prob_scale=prob_scale, blank_bias=blank_bias)
osd = vo.state_from_numpy(sd_np, requires_grad=False)
idx_to_char = {i: c for i, c in enumerate(chars)}
for seed in seeds:
    x, w, tgt, tl = cf.closed_form_batch(B, 1, 30, [600] * B, V, [20] * B, seed=seed)
    r = np.random.RandomState(121)          # pool samples and dropout masks as in the test: fixed, only the batch varies
    s1 = torch.from_numpy(r.uniform(0, 0.999, size=(B, 64, 2)).astype(np.float32))
    s2 = torch.from_numpy(r.uniform(0, 0.999, size=(B, 128, 2)).astype(np.float32))
    masks = [torch.from_numpy((r.uniform(size=(T, B, 2 * H)) >= 0.5).astype(np.float32) * 2.0) for _ in range(2)] if dropout else None
    t0 = time.time()
    with torch.no_grad():
        lo, ln = vo.forward(osd, hp, torch.from_numpy(x), w, (s1, s2), training=True, dropout_masks=masks,
                            lstm_training=None if dropout else False)
    top2 = torch.sort(lo, dim=2, descending=True)[0]
    margin = (top2[:, :, 0] - top2[:, :, 1])
    labels = vo.greedy_decode(lo, ln, idx_to_char, uxxxx=True)[1]
    print("seed %d: min margin %.3e, decode margin %.3e, frames < 1e-3: %d, labels emitted %d, |logit|max %.1f, %.1f s"
          % (seed, float(margin.min()), decode_margin(lo, ln, V), int((margin < 1e-3).sum()), sum(len(l) for l in labels), float(lo.abs().max()), time.time() - t0), flush=True)
