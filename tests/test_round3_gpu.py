"""GPU: round-3 cases.

  * data-parallel replicas with per-rank randomness: two real ranks (gloo on one GPU always; RCCL on two GPUs when the box has them)
    run the identical init, va.seed_rank, one forward of the SAME batch (logits must differ: the ranks' FractionalMaxPool samples
    differ) and two train steps on per-rank batches (weights must stay bit-identical across ranks);
  * the health words are report-only: a raised word does not poison later sweeps, and a report is cleared when it is raised;
  * FlatClampAdam refuses step() while a bucket of the exchange is in flight and nobody collected it."""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_ranks(n, backend, share_gpu, out, limit_s=900):
    """Fresh child process per rank (nothing is exec'ed from a GPU-initialised process); the first failing rank ends the others."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = {k: v for k, v in os.environ.items()}
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        if share_gpu:
            env["VOCR_LSTM_SWEEP"] = "step"         # two processes' persistent sweeps must not compete for one GPU's CUs
        cmd = [sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), "--backend", backend, "--out", out] + (["--share-gpu"] if share_gpu else [])
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    t0 = time.time()
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc == 0 for rc in rcs):
            break
        if any(rc not in (None, 0) for rc in rcs) or time.time() - t0 > limit_s:
            for p in procs:
                if p.poll() is None:
                    p.kill()
            raise AssertionError("ranks failed: %s\n%s" % (rcs, "\n----\n".join((p.communicate()[0] or "")[-1500:] for p in procs)))
        time.sleep(0.2)
    return json.load(open(out))


def _check_replicas(recs, n):
    assert len(recs) == n and sorted(r["rank"] for r in recs) == list(range(n)) and all(r["world"] == n for r in recs)
    assert len({r["init_hash"] for r in recs}) == 1, "replicas must start from the identical init"
    assert len({r["dropout_seed"] for r in recs}) == n, "every rank needs its own dropout stream"
    assert len({r["logit_sum"] for r in recs}) == n, "the same batch must give different logits on different ranks (own pool samples)"
    assert len({r["weights_hash"] for r in recs}) == 1, "after summed-gradient steps every rank must hold bit-identical weights"
    assert all(np.isfinite(l) for r in recs for l in r["losses"])
    assert len({tuple(r["losses"]) for r in recs}) == n, "per-rank batches: the ranks' losses differ"


def test_two_ranks_on_one_gpu_rank_rng_and_identical_weights(tmp_path):
    recs = _run_ranks(2, "gloo", True, os.path.join(tmp_path, "dp.json"))
    _check_replicas(recs, 2)


def test_two_ranks_on_two_gpus_over_rccl(tmp_path):
    """The real thing: one process per GPU, RCCL all-reduce in two buckets inside the backward, persistent sweeps on.  Skipped on the
    1-GPU boxes of this pool; the driver's multi-GPU bench is the first place it can run."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (this box has %d)" % torch.cuda.device_count())
    recs = _run_ranks(2, "nccl", False, os.path.join(tmp_path, "dp.json"))
    _check_replicas(recs, 2)
    assert {r["device"] for r in recs} == {0, 1}


def test_health_word_reports_but_never_poisons_later_sweeps():
    """the sweeps used to test the caller's sticky health word, so one transient hand-off timeout NaN-poisoned every later
    sweep of the process.  Now a raised word changes no result, check_health() raises once and clears the report."""
    import vistaocr_amd as va
    from vistaocr_amd import ops
    dev = torch.device("cuda", 0)
    al = va.english_alphabet()
    torch.manual_seed(0)
    model = va.CnnOcrModel(alphabet=al, verbose=False, input_line_height=30, rds_line_height=30, lstm_input_dim=32, num_lstm_layers=1,
                           num_lstm_hidden_units=256, p_lstm_dropout=0.0, num_in_channels=1)
    model.eval()
    x = torch.rand(3, 1, 30, 120)
    widths = torch.tensor([120, 100, 60], dtype=torch.int32)
    model.pool_samples = [torch.rand(3, 64, 2), torch.rand(3, 128, 2)]
    with torch.no_grad():
        ref, _ = model(x, widths)
    ops.health(dev)[0] = 1                           # as if an earlier sweep had timed out
    with torch.no_grad():
        again, _ = model(x, widths)
    assert torch.isfinite(again).all() and torch.equal(again, ref), "a stale report must not change a later sweep's result"
    with pytest.raises(RuntimeError, match="timed out"):
        ops.check_health_sync(dev)
    ops.check_health_sync(dev)                       # the report was cleared when it was raised
    assert ops.health(dev).cpu().tolist() == [0, 0]
    ops.health(dev)[1] = 1
    with pytest.raises(RuntimeError, match="NaN gradient"):
        ops.check_health_sync(dev)
    ops.check_health_sync(dev)


def test_step_refuses_an_uncollected_exchange():
    """backward() starts the sequence-side all-reduce when a process group is active; step() without all_reduce_grads() used
    to race it.  It now refuses (and zero_grad() waits for the collective before clearing the buffer)."""
    import torch.distributed as dist
    import vistaocr_amd as va
    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this interpreter")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 2000), VOCR_FORCE_DIST="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        torch.manual_seed(0)
        model = va.CnnOcrModel(alphabet=va.english_alphabet(), verbose=False, input_line_height=30, rds_line_height=30, lstm_input_dim=16,
                               num_lstm_layers=1, num_lstm_hidden_units=64, p_lstm_dropout=0.0)
        opt = va.make_optimizer(model)
        crit = va.CTCLoss()
        model.train()
        x = torch.rand(2, 1, 30, 100)
        widths = torch.tensor([100, 100], dtype=torch.int32)
        tgt = torch.tensor([3, 4, 5, 6], dtype=torch.int32)
        tl = torch.tensor([2, 2], dtype=torch.int32)
        opt.zero_grad()
        lg, lens = model(x, widths)
        crit(lg, tgt, lens, tl).backward()
        assert opt._tail_work is not None
        with pytest.raises(RuntimeError, match="all_reduce_grads"):
            opt.step()
        assert opt._tail_work is None
        opt.zero_grad()
        lg, lens = model(x, widths)
        crit(lg, tgt, lens, tl).backward()
        opt.all_reduce_grads()
        opt.step()
        opt.check_health()
    finally:
        os.environ.pop("VOCR_FORCE_DIST", None)
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------- panel GEMM (gemm_dma.hip)
def _r(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * 2 - 1


def _check(c, ref, k, what):
    err = (c.detach().cpu().double() - ref).abs()
    tol = 3e-6 * k + 1e-5 * ref.abs()
    assert bool((err <= tol).all()), "%s: max abs err %.3e (scale %.2e)" % (what, float(err.max()), float(ref.abs().max()))


@pytest.mark.parametrize("m,n,k", [(9408, 2048, 1024), (9408, 1024, 2048), (2048, 1024, 9408), (4200, 1000, 1000), (2052, 512, 4096),
                                   (9408, 2048, 128), (2560, 384, 8192), (8200, 260, 72), (2080, 128, 4096)])
def test_panel_gemm_all_layouts(m, n, k):
    """The DMA-staged panel kernel on the step's large products and on edge shapes (M, N, K not multiples of the tile sizes, K cut into
    slabs for the long-K / few-tile ones), all four operand layouts, bias + ReLU, accumulate; bitwise reproducible."""
    from vistaocr_amd import ops
    dev = torch.device("cuda", 0)
    a, b, bias = _r((m, k), 1), _r((k, n), 2), _r((n,), 3)
    ref = a.double() @ b.double()
    ad, bd, at, bt = a.to(dev), b.to(dev), a.t().contiguous().to(dev), b.t().contiguous().to(dev)
    c = torch.empty(m, n, device=dev)
    ops.gemm(0, 1, m, n, k, ad, k, bt, k, c, n, bias=bias.to(dev), relu=True)
    _check(c, torch.relu(ref + bias.double()), k, "NT bias relu")
    c1 = torch.empty(m, n, device=dev)
    ops.gemm(0, 1, m, n, k, ad, k, bt, k, c1, n, bias=bias.to(dev), relu=True)
    assert torch.equal(c, c1), "two runs must agree bit for bit"
    ops.gemm(0, 0, m, n, k, ad, k, bd, n, c, n)
    _check(c, ref, k, "NN")
    ops.gemm(1, 0, m, n, k, at, m, bd, n, c, n)
    _check(c, ref, k, "TN")
    c2 = torch.empty(m, n, device=dev)
    ops.gemm(1, 0, m, n, k, at, m, bd, n, c2, n)
    assert torch.equal(c, c2), "two runs (K slabs) must agree bit for bit"
    ops.gemm(1, 1, m, n, k, at, m, bt, k, c, n, bias=bias.to(dev))
    _check(c, ref + bias.double(), k, "TT bias")
    c0 = _r((m, n), 4)
    c = c0.clone().to(dev)
    ops.gemm(1, 0, m, n, k, at, m, bd, n, c, n, accumulate=True)
    _check(c, ref + c0.double(), k, "TN accumulate")
    # output and operands embedded in wider matrices (leading dimensions > extents), like the LSTM's y[:, H:] / xproj[d] views
    wide = torch.zeros(m, n + 64, device=dev)
    awide = torch.zeros(m, k + 32, device=dev)
    awide[:, 16:16 + k] = ad
    ops.gemm(0, 0, m, n, k, awide[:, 16:], k + 32, bd, n, wide[:, 32:], n + 64)
    _check(wide[:, 32:32 + n], ref, k, "NN with leading dimensions")
    assert float(wide[:, :32].abs().max()) == 0.0 and float(wide[:, 32 + n:].abs().max()) == 0.0, "wrote outside the output view"


@pytest.mark.parametrize("m,n,k", [(9408, 2048, 1024), (9408, 2048, 128), (2048, 1024, 9408), (2048, 512, 9376), (9408, 1024, 2048),
                                   (9408, 128, 2048), (2048, 128, 9408), (1000, 300, 520)])
def test_gemm_pair_modes(m, n, k):
    """vocr_gemm_pair: two products of one shape in one launch (mode 0) and one product whose K runs through two operand pairs (mode 1)
    against fp64; shapes the panel kernel does not take fall back to two vocr_gemm calls with the same results contract."""
    from vistaocr_amd import ops
    dev = torch.device("cuda", 0)
    a0, a1, b0, b1 = _r((m, k), 1), _r((m, k), 2), _r((k, n), 3), _r((k, n), 4)
    bias0, bias1 = _r((n,), 5), _r((n,), 6)
    r0, r1 = a0.double() @ b0.double(), a1.double() @ b1.double()
    d = lambda t: t.to(dev)
    c0, c1 = torch.empty(m, n, device=dev), torch.empty(m, n, device=dev)
    # mode 0, NT with biases (the x-projections: one A, two weight matrices)
    ops.gemm_pair(0, 0, 1, m, n, k, d(a0), d(a0), k, d(b0.t().contiguous()), d(b1.t().contiguous()), k, c0, c1, n, bias0=d(bias0), bias1=d(bias1))
    _check(c0, r0 + bias0.double(), k, "pair mode 0 NT / product 0")
    _check(c1, a0.double() @ b1.double() + bias1.double(), k, "pair mode 0 NT / product 1")
    # mode 0, TN (the weight gradients: two A, one B)
    ops.gemm_pair(0, 1, 0, m, n, k, d(a0.t().contiguous()), d(a1.t().contiguous()), m, d(b0), d(b0), n, c0, c1, n)
    _check(c0, r0, k, "pair mode 0 TN / product 0")
    _check(c1, a1.double() @ b0.double(), k, "pair mode 0 TN / product 1")
    e0, e1 = torch.empty(m, n, device=dev), torch.empty(m, n, device=dev)
    ops.gemm_pair(0, 1, 0, m, n, k, d(a0.t().contiguous()), d(a1.t().contiguous()), m, d(b0), d(b0), n, e0, e1, n)
    assert torch.equal(c0, e0) and torch.equal(c1, e1), "two runs must agree bit for bit"
    # mode 1, NN (the data gradient: K through both pairs)
    ops.gemm_pair(1, 0, 0, m, n, k, d(a0), d(a1), k, d(b0), d(b1), n, c0, None, n)
    _check(c0, r0 + r1, 2 * k, "pair mode 1 NN")
    ops.gemm_pair(1, 0, 0, m, n, k, d(a0), d(a1), k, d(b0), d(b1), n, e0, None, n, bias0=d(bias0), relu=True)
    _check(e0, torch.relu(r0 + r1 + bias0.double()), 2 * k, "pair mode 1 NN bias relu")


def test_a_process_group_does_not_cost_the_step_its_overlap():
    """Regression guard for the hardware-queue trap: with a process group initialised (RCCL, one rank) the step must take
    what it takes without one.  An extra stream created on the data-parallel path once cost 3.4 of 18.5 ms; the bound is generous."""
    def run(extra_env):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        env.update(extra_env)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-1500:]
        return json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    plain = run({})
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist = run({"VOCR_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1"})
    assert dist["config"]["backend"].startswith("nccl") and dist["allreduce_ms_per_step"] is not None
    print("ms per step: %.2f without a process group, %.2f with one (RCCL, one rank)" % (plain["ms_per_step"], dist["ms_per_step"]))
    assert dist["ms_per_step"] <= 1.06 * plain["ms_per_step"], (plain["ms_per_step"], dist["ms_per_step"])


def test_profile_ranges_are_harmless_and_balanced():
    """vocr_profile_range_push / pop (roctx, dlopen'ed lazily): 0 = recorded, 1 = roctx not on this machine; the Python mirror's ranges
    (VOCR_ROCTX=1) must not change a result."""
    from vistaocr_amd import _lib
    lib = _lib.load()
    rc = lib.vocr_profile_range_push(b"test.range")
    assert rc in (0, 1)
    assert lib.vocr_profile_range_pop() == rc
    assert lib.vocr_profile_range_push(None) < 0
    env = dict(os.environ, VOCR_ROCTX="1")
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stderr[-1500:]


@pytest.mark.parametrize("T,B,H", [(21, 32, 512), (17, 9, 512), (13, 30, 256), (9, 40, 128)])
def test_dropout_inside_the_layer_op_equals_the_separate_op(T, B, H):
    """BiLstmLayerFn with its output dropout drawn inside - with the backward mask as a multiply of its own, and (VOCR_LSTM_BWD_PARTS=1)
    riding on the sweep's read of dy with the bias gradient's last reduction beside the weight gradients (vocr_lstm_bwd_parts /
    vocr_lstm_bias_from_parts) - == the plain layer followed by DropoutFn: same mask function, same products - bit for bit, outputs
    and every gradient."""
    from vistaocr_amd import ops
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    D = 64
    x0 = ((torch.rand(T * B, D, generator=g) - 0.5)).to(dev)
    lens = torch.tensor(sorted([max(1, T - (i * T) // (B + 2)) for i in range(B)], reverse=True), dtype=torch.int32, device=dev)
    shapes = [(4 * H, D), (4 * H, H), (4 * H,), (4 * H,)] * 2
    params0 = [((torch.rand(*sh, generator=g) - 0.5) * 0.2).to(dev) for sh in shapes]
    dout = ((torch.rand(T * B, 2 * H, generator=g) - 0.5)).to(dev)
    res = []
    for fused in (True, "parts", False):
        os.environ["VOCR_LSTM_BWD_PARTS"] = "1" if fused == "parts" else "0"
        os.environ["VOCR_EXPERIMENTS"] = "1"       # experiment switches are read only then (vistaocr_amd/ops.py:_exp)
        x = x0.clone().requires_grad_(True)
        ps = [p.clone().requires_grad_(True) for p in params0]
        if fused:
            out = ops.BiLstmLayerFn.apply(x, lens, T, B, *ps, None, True, 0.5, 1234)
        else:
            out = ops.DropoutFn.apply(ops.BiLstmLayerFn.apply(x, lens, T, B, *ps), 0.5, 1234)
        out.backward(dout)
        ops.join_side_stream(dev)
        torch.cuda.synchronize()
        res.append([out.detach().cpu(), x.grad.cpu()] + [p.grad.cpu() for p in ps])
    os.environ.pop("VOCR_LSTM_BWD_PARTS", None)
    os.environ.pop("VOCR_EXPERIMENTS", None)
    ops.check_health_sync(dev)
    names = ["out", "dx", "dW_ih", "dW_hh", "db_ih", "db_hh", "dW_ih_r", "dW_hh_r", "db_ih_r", "db_hh_r"]
    for other in res[:2]:
        for nm, a, b in zip(names, other, res[2]):
            assert torch.equal(a, b), "%s differs: max |diff| %.3e" % (nm, float((a - b).abs().max()))
    assert float((res[0][0] == 0).float().mean()) > 0.3          # the mask did drop


@pytest.mark.parametrize("B,H", [(32, 512), (12, 512), (27, 256)])
def test_a_hand_off_that_never_arrives_ends_loudly_not_in_a_hang(B, H):
    """The self-validating sweeps poll with bounded spins.  Resuming a forward sweep at step 3 on a workspace whose hand-off ring was
    never written (all "not written yet") is a hand-off that cannot arrive: the call must come back (seconds), raise the caller's
    health word and poison the output of its last step with NaN - and the next, proper sweep on the same workspace must be clean."""
    from vistaocr_amd import _lib
    from vistaocr_amd._lib import call
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    T = 6
    g = torch.Generator().manual_seed(3)
    xp = (torch.rand(2, T * B, 4 * H, generator=g) - 0.5).to(dev)
    wf = ((torch.rand(4 * H, H, generator=g) - 0.5) * 0.2).to(dev)
    wr = ((torch.rand(4 * H, H, generator=g) - 0.5) * 0.2).to(dev)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    y = torch.zeros(T * B, 2 * H, device=dev)
    gt = torch.zeros(2, T * B, 4 * H, device=dev)
    c = torch.zeros(2, T * B, H, device=dev)
    ws = torch.full((lib.vocr_lstm_workspace_bytes(T, B, H) // 4 + 16,), float("nan"), device=dev)
    ws.view(torch.int32).fill_(-1)                                  # every ring slot: "not written yet"
    hw = torch.zeros(4, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    t0 = time.time()
    call("vocr_lstm_fwd_range", xp.data_ptr(), wf.data_ptr(), wr.data_ptr(), lens.data_ptr(), y.data_ptr(), gt.data_ptr(), c.data_ptr(),
         ws.data_ptr(), T, B, H, 3, T, hw.data_ptr(), s)
    torch.cuda.synchronize()
    assert time.time() - t0 < 60, "the bounded spins took %.0f s" % (time.time() - t0)
    assert int(hw[0]) == 1, "the time-out was not reported"
    y3 = y.view(T, B, 2, H)
    assert torch.isnan(y3[T - 1, :, 0]).any() and torch.isnan(y3[0, :, 1]).any(), "the last step of both directions must be poisoned"
    hw.zero_()
    call("vocr_lstm_fwd", xp.data_ptr(), wf.data_ptr(), wr.data_ptr(), lens.data_ptr(), y.data_ptr(), gt.data_ptr(), c.data_ptr(),
         ws.data_ptr(), T, B, H, hw.data_ptr(), s)
    torch.cuda.synchronize()
    assert int(hw[0]) == 0 and torch.isfinite(y).all(), "a proper sweep after a failed one must be clean"
