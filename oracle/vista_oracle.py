"""TEST INFRASTRUCTURE ONLY — CPU restatement of VistaOCR's CnnOcrModel hot path on PyTorch-CPU.

Parity status: PINNED.  Every function here is checked by tests/test_oracle_golden.py against the
golden vectors in tests/golden/*.npz, which oracle/gen_golden.py produced in the build container by
importing the real reference (`/root/reference/src/models/cnnlstm.py`) on the PyTorch-CPU path.
The one piece of the path whose arithmetic is NOT in the reference tree is the CTC criterion:
`warpctc_pytorch.CTCLoss` (SeanNaren/warp-ctc binding of baidu-research/warp-ctc; no version is
pinned anywhere in the reference — no requirements file, lock file or submodule).  Per
BASELINE.json's north_star ("match the reference PyTorch-CPU path") its restatement is
torch.nn.functional.ctc_loss(log_softmax(logits), reduction='sum') (same definition as warp-ctc's
batch-summed cost with internal softmax); oracle/ctc_ref.c restates the published alpha/beta
algorithm in plain C and is checked against it.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference map (all paths under /root/reference/):
  forward()               src/models/cnnlstm.py:268-296
  conv/bn/relu/pool plan  src/models/cnnlstm.py:114-134, 263-266
  bridge / lstm / prob    src/models/cnnlstm.py:143-154, 275-294
  output_width()          src/models/cnnlstm.py:211-260
  ctc_criterion()         src/train_cnn_lstm.py:358,138 (warpctc_pytorch.CTCLoss call sites)
  greedy_decode()         src/models/cnnlstm.py:479-541 == src/decoder.py:116-185
  train_step()            src/train_cnn_lstm.py:131-150
  uxxxx_to_utf8()         src/textutils.py:216-243
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

CONV_PLAN = (0, 3, "pool", 7, 10, "pool", 14, 17, 20)   # nn.Sequential indices, cnnlstm.py:124-134


# ----------------------------------------------------------------------------- size bookkeeping
def num_rds_layers(hp):
    """cnnlstm.py:97-112 (power-of-two check; note the reference divides with `/=`)."""
    ih, rh = hp["input_line_height"], hp["rds_line_height"]
    if rh > ih:
        raise Exception("rapid-downsample line height must be less than or equal to input line height")
    if ih % rh != 0:
        raise Exception("rapid-downsample line height must evenly divide input line height by a power of 2")
    n, lh = 0, ih
    while lh > rh:
        n += 1
        if lh % 2 != 0:
            raise Exception("rapid-downsample line height must eenly diide input line height by a power of 2")
        lh /= 2
    if lh != rh:
        raise Exception("rapid-downsample line height must eenly diide input line height by a power of 2")
    return n


def output_size(hp, h, w):
    """cnn_input_size_to_output_size (cnnlstm.py:211-260): conv k3 p1 keeps size, MaxPool2d(2,2)
    halves with floor in double, FractionalMaxPool2d -> floor(h*0.5), floor(w*0.7) in double."""
    for _ in range(num_rds_layers(hp)):
        h = math.floor((h + 2.0 * 1 - 1 * (3 - 1) - 1) / 1 + 1)
        w = math.floor((w + 2.0 * 1 - 1 * (3 - 1) - 1) / 1 + 1)
        h = math.floor((h + 2.0 * 0 - 1 * (2 - 1) - 1) / 2 + 1)
        w = math.floor((w + 2.0 * 0 - 1 * (2 - 1) - 1) / 2 + 1)
    for step in CONV_PLAN:
        if step == "pool":
            h, w = math.floor(h * 0.5), math.floor(w * 0.7)
        else:
            h = math.floor((h + 2.0 - 2 - 1) / 1 + 1)
            w = math.floor((w + 2.0 - 2 - 1) / 1 + 1)
    return h, w


def fracpool_intervals(sample, in_size, out_size, pool=2):
    """ATen fractional_max_pool2d interval rule restated in numpy float32:
    alpha = f32(in-pool)/f32(out-1); start[i] = int((i+u)*alpha) - int(u*alpha); last = in-pool."""
    seq = np.zeros(out_size, dtype=np.int64)
    u = np.float32(sample)
    if out_size > 1:
        alpha = np.float32(in_size - pool) / np.float32(out_size - 1)
        for i in range(out_size - 1):
            seq[i] = int(np.float32(np.float32(i) + u) * alpha) - int(u * alpha)
    if out_size > 0:
        seq[out_size - 1] = in_size - pool
    return seq


def fracpool_numpy(x, samples, out_h, out_w):
    """Pure-numpy fractional 2x2 max pool (first max wins, flat index into the H*W plane)."""
    N, C, H, W = x.shape
    out = np.empty((N, C, out_h, out_w), dtype=x.dtype)
    idx = np.empty((N, C, out_h, out_w), dtype=np.int64)
    for n in range(N):
        for c in range(C):
            sw = fracpool_intervals(samples[n, c, 0], W, out_w)
            sh = fracpool_intervals(samples[n, c, 1], H, out_h)
            p = x[n, c]
            cand = np.stack([p[sh[:, None] + dh, sw[None, :] + dw] for dh in (0, 1) for dw in (0, 1)], 0)
            k = cand.argmax(0)            # first max wins, window scanned row-major like ATen
            out[n, c] = np.take_along_axis(cand, k[None], 0)[0]
            idx[n, c] = (sh[:, None] + k // 2) * W + (sw[None, :] + k % 2)
    return out, idx


# ----------------------------------------------------------------------------- parameters
def state_from_numpy(sd_np, requires_grad=True):
    sd = {}
    for k, v in sd_np.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).clone()
        if requires_grad and not (k.endswith("running_mean") or k.endswith("running_var")):
            t.requires_grad_(True)
        sd[k] = t
    return sd


def init_uniform_state(hp, vocab, seed=0):
    """cnnlstm.py:158-159: uniform(-0.08, 0.08) on EVERY parameter (BN gamma/beta, biases too)."""
    from oracle.closed_form import param_shapes
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(hp, vocab):
        if name.endswith("running_mean"):
            sd[name] = torch.zeros(shape)
        elif name.endswith("running_var"):
            sd[name] = torch.ones(shape)
        else:
            sd[name] = (torch.rand(shape, generator=g) * 0.16 - 0.08).requires_grad_(True)
    return sd


def trainable(sd):
    return [(k, v) for k, v in sd.items() if v.requires_grad]


# ----------------------------------------------------------------------------- conv operand rounding (config 5)
class _ConvF16Operands(torch.autograd.Function):
    """Semantics of the build's fp16-operand conv (BASELINE config 5): forward, data-gradient and weight-gradient all
    multiply fp16-ROUNDED operands and accumulate in fp32 (first layer, Cin <= 3: weight gradient from fp32 operands)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return F.conv2d(x.half().float(), w.half().float(), b, padding=1)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = torch.nn.grad.conv2d_input(x.shape, w.half().float(), dy.half().float(), padding=1)
        if x.shape[1] <= 3:
            dw = torch.nn.grad.conv2d_weight(x, w.shape, dy, padding=1)
        else:
            dw = torch.nn.grad.conv2d_weight(x.half().float(), w.shape, dy.half().float(), padding=1)
        return dx, dw, dy.sum((0, 2, 3))


def _conv(x, w, b, conv_dtype):
    if conv_dtype == "fp16":
        return _ConvF16Operands.apply(x, w, b)
    return F.conv2d(x, w, b, padding=1)


# ----------------------------------------------------------------------------- forward
def _lstm_module(sd, hp):
    D, H, L = hp["lstm_input_dim"], hp["num_lstm_hidden_units"], hp["num_lstm_layers"]
    m = torch.nn.LSTM(D, H, num_layers=L, dropout=hp["p_lstm_dropout"], bidirectional=True)
    # share storage with the oracle's parameter dict so autograd reaches sd[...]
    for name in list(m._parameters.keys()):
        del m._parameters[name]
        setattr(m, name, sd["lstm." + name])
    m._flat_weights = [getattr(m, n) for n in m._flat_weights_names]
    return m


def lstm_explicit(x, lens, sd, hp, dropout_masks=None):
    """Explicit time-step restatement of nn.LSTM on a packed batch (gate order i,f,g,o; both bias
    vectors; reverse direction starts at each sequence's own last frame; padded outputs are zero).
    dropout_masks: optional list (per non-final layer) of [T,B,2H] multiplicative masks that already
    include the 1/(1-p) scale — the explicit-mask contract of the build."""
    T, B, _ = x.shape
    H, L = hp["num_lstm_hidden_units"], hp["num_lstm_layers"]
    lens = [int(v) for v in lens]
    inp = x
    for l in range(L):
        outs = []
        for sfx, order in (("", range(T)), ("_reverse", range(T - 1, -1, -1))):
            w_ih, w_hh = sd["lstm.weight_ih_l%d%s" % (l, sfx)], sd["lstm.weight_hh_l%d%s" % (l, sfx)]
            bias = sd["lstm.bias_ih_l%d%s" % (l, sfx)] + sd["lstm.bias_hh_l%d%s" % (l, sfx)]
            h = x.new_zeros(B, H)
            c = x.new_zeros(B, H)
            ys = [None] * T
            for t in order:
                act = torch.tensor([1.0 if t < lens[b] else 0.0 for b in range(B)]).unsqueeze(1)
                g = inp[t] @ w_ih.t() + h @ w_hh.t() + bias
                i, f, gg, o = g.chunk(4, 1)
                c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
                h_new = torch.sigmoid(o) * torch.tanh(c_new)
                c = act * c_new + (1 - act) * c
                h = act * h_new + (1 - act) * h
                ys[t] = act * h_new
            outs.append(torch.stack(ys, 0))
        inp = torch.cat(outs, 2)
        if dropout_masks is not None and l < L - 1:
            inp = inp * dropout_masks[l]
    return inp


def forward(sd, hp, x, widths, pool_samples, training, lstm_training=None, dropout_masks=None,
            taps=None):
    """CnnOcrModel.forward (cnnlstm.py:268-296) -> (logits[T,B,V], lens int32[B]).
    `pool_samples` = (u1[B,64,2], u2[B,128,2]) injected as FractionalMaxPool2d._random_samples.
    `training` drives BatchNorm (batch statistics + running-stat update in place on sd);
    `lstm_training` (default = training) drives nn.LSTM's internal dropout; an explicit
    `dropout_masks` list switches the LSTM to lstm_explicit()."""
    if lstm_training is None:
        lstm_training = training
    a = x
    for i in range(num_rds_layers(hp)):                              # cnnlstm.py:114-121
        a = _conv(a, sd["rapid_ds.%02d-conv.weight" % i], sd["rapid_ds.%02d-conv.bias" % i], hp.get("conv_dtype", "fp32"))
        a = F.max_pool2d(F.relu(a), 2, stride=2)
    pool_i = 0
    for step in CONV_PLAN:                                           # cnnlstm.py:124-134
        if step == "pool":
            oh, ow = math.floor(a.shape[2] * 0.5), math.floor(a.shape[3] * 0.7)
            a = F.fractional_max_pool2d(a, 2, output_size=(oh, ow), _random_samples=pool_samples[pool_i])
            if taps is not None:
                taps["pool%d" % pool_i] = a
            pool_i += 1
            continue
        a = _conv(a, sd["cnn.%d.weight" % step], sd["cnn.%d.bias" % step], hp.get("conv_dtype", "fp32"))
        bn = "cnn.%d." % (step + 1)
        a = F.batch_norm(a, sd[bn + "running_mean"], sd[bn + "running_var"], sd[bn + "weight"], sd[bn + "bias"],
                         training=training, momentum=0.1, eps=1e-5)
        a = F.relu(a)
        if taps is not None:
            taps["act%d" % step] = a
    b, c, h, w = a.shape                                             # cnnlstm.py:275-278
    feat = a.permute(3, 0, 1, 2).contiguous().view(-1, c * h)
    lstm_in = F.relu(F.linear(feat, sd["bridge_layer.0.weight"], sd["bridge_layer.0.bias"])).view(w, b, -1)
    if taps is not None:
        taps["bridge"] = lstm_in
    out_w = [output_size(hp, hp["input_line_height"], int(wd))[1] for wd in widths]   # :285-286
    if dropout_masks is not None:
        T = max(out_w)
        lstm_out = lstm_explicit(lstm_in[:T], out_w, sd, hp, dropout_masks)
        lens = torch.tensor(out_w, dtype=torch.int64)
    else:
        m = _lstm_module(sd, hp)
        m.train(lstm_training)
        packed = pack_padded_sequence(lstm_in, out_w)                # requires descending widths
        packed_out, _ = m(packed)
        lstm_out, lens = pad_packed_sequence(packed_out)
    if taps is not None:
        taps["lstm"] = lstm_out
    T = lstm_out.shape[0]
    logits = F.linear(lstm_out.reshape(-1, lstm_out.shape[2]), sd["prob_layer.0.weight"],
                      sd["prob_layer.0.bias"]).view(T, b, -1)
    return logits, lens.to(torch.int32)


# ----------------------------------------------------------------------------- criterion
def ctc_criterion(logits, targets, act_lens, target_lens):
    """warpctc_pytorch.CTCLoss() call contract (train_cnn_lstm.py:138): pre-softmax logits [T,B,V],
    flat int32 CPU targets, int32 CPU lengths -> Tensor shape (1,) = batch-SUMMED NLL, blank = 0."""
    lp = F.log_softmax(logits, dim=2)
    loss = F.ctc_loss(lp, targets.to(torch.int64), act_lens.to(torch.int64), target_lens.to(torch.int64),
                      blank=0, reduction="sum", zero_infinity=False)
    return loss.reshape(1)


# ----------------------------------------------------------------------------- decode
def uxxxx_to_utf8(in_str):
    """textutils.py:216-243."""
    if in_str.strip() == "":
        return ""
    out = ""
    for tok in in_str.split():
        if tok == "":
            continue
        if tok in ("<unk>", "<s>", "</s>"):
            out += tok
        else:
            out += chr(int(tok[1:], 16))
    return out


def greedy_decode(logits, lens, idx_to_char, uxxxx=False):
    """decode_without_lm (cnnlstm.py:479-541).  Returns (strings, integer label sequences).
    Threshold 3/len(alphabet) is applied to RAW logits; repeats collapse by comparing the
    alphabet STRINGS (so two indices with the same uxxxx code collapse together)."""
    V = len(idx_to_char)
    thresh = 3 * 1 / V
    T, B = logits.shape[0], logits.shape[1]
    prev = ["" for _ in range(B)]
    res = ["" for _ in range(B)]
    labels = [[] for _ in range(B)]
    for t in range(T):
        row = logits.data[t].cpu().numpy()
        mx = row.max(1).flatten()
        am = row.argmax(1).flatten()
        for b in range(B):
            if t >= lens[b]:
                continue
            if am[b] == 0:
                prev[b] = ""
                continue
            if mx[b] < thresh:
                prev[b] = ""
                continue
            ch = idx_to_char[int(am[b])]
            if prev[b] == ch:
                continue
            res[b] += ch
            labels[b].append(int(am[b]))
            prev[b] = ch
            if t != T - 1:
                res[b] += " "
    for b in range(B):
        if len(res[b]) > 0 and res[b][-1] == " ":
            res[b] = res[b][:-1]
    if not uxxxx:
        res = [uxxxx_to_utf8(r) for r in res]
    return res, labels


# ----------------------------------------------------------------------------- train step
def train_step(sd, hp, optimizer, x, widths, targets, target_lens, pool_samples, lstm_training=True,
               dropout_masks=None):
    """train() (train_cnn_lstm.py:131-150): zero_grad, forward, summed CTC loss, backward,
    elementwise clamp of every gradient to [-5, 5], optimizer.step(); returns the loss float."""
    optimizer.zero_grad()
    logits, lens = forward(sd, hp, x, widths, pool_samples, training=True, lstm_training=lstm_training,
                           dropout_masks=dropout_masks)
    loss = ctc_criterion(logits, targets, lens, target_lens)
    loss.backward()
    for _, p in trainable(sd):
        if p.grad is not None:
            p.grad.data.clamp_(min=-5, max=5)
    optimizer.step()
    return float(loss.item()), logits.detach(), lens
