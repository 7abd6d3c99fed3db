"""Shared checks of the GPU parity tests (HIP path vs the on-box oracle).

  * decode_margin(): how far the ORACLE's greedy decode is from changing its mind on any valid frame - the top-2 logit gap and,
    for non-blank frames, the distance of the maximum from the reference's raw-logit threshold 3/len(alphabet)
    (src/models/cnnlstm.py:481,515).  Tests pick their batch seed on the oracle side until this margin is comfortable
    (pick_seed), so that "greedy label sequences bit-exact" can be asserted unconditionally and is not vacuous.
  * assert_grads_close(): every gradient tensor ELEMENT-WISE, ||g_hip - g_oracle||_2 / ||g_oracle||_2 <= rtol (a norm-only
    comparison cannot see a transposed, permuted or sign-flipped block)."""
import torch

# conv biases in front of a batch-statistics BatchNorm: exactly zero gradient in exact arithmetic, rounding noise on both sides
_BN_SHADOWED_BIAS = tuple("cnn.%d.bias" % i for i in (0, 3, 7, 10, 14, 17, 20))


def decode_margin(logits, lens, vocab):
    """min over valid frames of min(top-1 - top-2, |top-1 - 3/vocab| on frames whose argmax is not the blank)."""
    T = logits.shape[0]
    lens = torch.as_tensor(lens).to(torch.int64)
    valid = torch.arange(T).unsqueeze(1) < lens.unsqueeze(0)
    top2 = torch.topk(logits.detach(), 2, dim=2)
    gap = (top2.values[:, :, 0] - top2.values[:, :, 1])[valid]
    nonblank = (top2.indices[:, :, 0] != 0)[valid]
    thr = (top2.values[:, :, 0][valid] - 3.0 / vocab).abs()[nonblank]
    m = float(gap.min())
    if thr.numel():
        m = min(m, float(thr.min()))
    return m


def pick_seed(make_logits, first_seed, want, tries=24):
    """Run `make_logits(seed) -> (logits, lens, vocab)` on the oracle for seed = first_seed, first_seed + 1, ... until
    decode_margin >= want; returns (seed, margin, result of make_logits).  `first_seed` is chosen offline so that the first try
    succeeds (scripts/margin_search.py); the loop only guards against a host CPU whose rounding differs."""
    best = None
    for seed in range(first_seed, first_seed + tries):
        res = make_logits(seed)
        m = decode_margin(res[0], res[1], res[2])
        if best is None or m > best[1]:
            best = (seed, m, res)
        if m >= want:
            return seed, m, res
    raise AssertionError("no batch seed in [%d, %d) gives the oracle a greedy-decode margin >= %.1e (best %.2e at seed %d)"
                         % (first_seed, first_seed + tries, want, best[1], best[0]))


def grad_rel_errors(named_hip_grads, oracle_state, skip_bn_shadowed_bias=True):
    """{name: ||g_hip - g_oracle|| / ||g_oracle||} in float64."""
    out = {}
    for k, g in named_hip_grads:
        if skip_bn_shadowed_bias and k in _BN_SHADOWED_BIAS:
            continue
        ref = oracle_state[k].grad.double()
        d = (g.detach().cpu().double() - ref).norm()
        out[k] = float(d / (ref.norm() + 1e-30)), float(ref.norm())
    return out


# fp32 against fp32 agrees to 1e-5 .. 1e-4 per kernel (tests/test_ops_gpu.py holds gemm_pair to 2e-5 of fp64 on the step's own views), but
# a weight gradient sums T x B outer products behind up to 588 recurrent steps, each side with its own summation order: measured
# HIP-vs-oracle up to 1.0e-3 on lstm.weight_ih_l1 of the ragged T = 588 case.  Every gradient that the GEMM, LSTM and CTC kernels produce
# (prob layer, LSTM, bridge) is held to 3e-3, so a systematic percent-level error in one of those kernels (a dropped K tail, a missing
# slab) fails.  The conv stack sits behind seven batch-statistics BatchNorms whose backward divides by sigma and subtracts two
# near-equal sums: measured HIP-vs-oracle 5e-5 at prob_layer growing to a few 1e-3 at cnn.0 (scripts/diag_golden.py), and two fp32
# CPU builds differ by as much; those tensors keep 1e-2 (their kernels are held to 1e-4 directly in tests/test_ops_gpu.py).
_RTOL_BY_PREFIX = (("cnn.", 1e-2), ("rapid_ds", 1e-2), ("", 3e-3))


def _rtol_for(name, override):
    if override is not None:
        return override
    for prefix, tol in _RTOL_BY_PREFIX:
        if name.startswith(prefix):
            return tol
    return 1e-2


def assert_grads_close(model, oracle_state, rtol=None, atol_norm=1e-5):
    """Element-wise gradient comparison of every parameter (rtol=None: the per-family tolerances above; a number: that tolerance for
    every tensor - the fp16-operand configuration's looser bar); returns (worst name, worst relative error)."""
    errs = grad_rel_errors(((k, p.grad) for k, p in model.named_parameters()), oracle_state)
    worst = ("", 0.0)
    for k, (rel, rn) in errs.items():
        if rel > worst[1]:
            worst = (k, rel)
        tol = _rtol_for(k, rtol)
        assert rel * rn <= tol * rn + atol_norm, "gradient of %s differs element-wise: rel L2 error %.3e > %.0e (|g| = %.3e)" % (k, rel, tol, rn)
    return worst
