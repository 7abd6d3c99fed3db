"""TEST INFRASTRUCTURE ONLY — closed-form deterministic tensors shared by the golden generator
(oracle/gen_golden.py, run once in the build container against the imported reference) and by the
tests that replay those goldens on the GPU box.  No file I/O, no reference import.

Tensors are drawn from numpy's legacy MT19937 `RandomState` (a stream numpy guarantees never to
change) keyed by crc32 of the state-dict key, so fixtures do not have to carry megabytes of
parameters: generator and tests rebuild bit-identical float32 tensors from (shape, name, scale).
"""
import zlib
import numpy as np


def _rs(name):
    """Legacy MT19937 RandomState keyed by crc32(name): numpy freezes this stream across versions."""
    return np.random.RandomState(zlib.crc32(name.encode("utf-8")) % (2 ** 31))


def fill(shape, name, scale=0.08, offset=0.0):
    return (offset + _rs(str(name)).uniform(-scale, scale, size=shape)).astype(np.float32)


def param_shapes(hp, vocab):
    """State-dict entries of CnnOcrModel in the reference's own key names and order
    (reference: src/models/cnnlstm.py:114-154; key names probed in SURVEY.md §8b)."""
    shapes = []
    cin = hp.get("num_in_channels", 1)
    lh = hp["input_line_height"]
    i = 0
    while lh > hp["rds_line_height"]:
        shapes.append(("rapid_ds.%02d-conv.weight" % i, (16, cin, 3, 3)))
        shapes.append(("rapid_ds.%02d-conv.bias" % i, (16,)))
        cin = 16
        lh //= 2
        i += 1
    plan = [(0, cin, 64), (3, 64, 64), (7, 64, 128), (10, 128, 128), (14, 128, 256), (17, 256, 256), (20, 256, 256)]
    for idx, ci, co in plan:
        shapes.append(("cnn.%d.weight" % idx, (co, ci, 3, 3)))
        shapes.append(("cnn.%d.bias" % idx, (co,)))
        shapes.append(("cnn.%d.weight" % (idx + 1), (co,)))
        shapes.append(("cnn.%d.bias" % (idx + 1), (co,)))
        shapes.append(("cnn.%d.running_mean" % (idx + 1), (co,)))
        shapes.append(("cnn.%d.running_var" % (idx + 1), (co,)))
    h_out = hp["rds_line_height"]
    h_out = int(np.floor(np.floor(h_out * 0.5) * 0.5))
    feat = 256 * h_out
    D = hp["lstm_input_dim"]
    H = hp["num_lstm_hidden_units"]
    shapes.append(("bridge_layer.0.weight", (D, feat)))
    shapes.append(("bridge_layer.0.bias", (D,)))
    for l in range(hp["num_lstm_layers"]):
        fin = D if l == 0 else 2 * H
        for sfx in ("", "_reverse"):
            shapes.append(("lstm.weight_ih_l%d%s" % (l, sfx), (4 * H, fin)))
            shapes.append(("lstm.weight_hh_l%d%s" % (l, sfx), (4 * H, H)))
            shapes.append(("lstm.bias_ih_l%d%s" % (l, sfx), (4 * H,)))
            shapes.append(("lstm.bias_hh_l%d%s" % (l, sfx), (4 * H,)))
    shapes.append(("prob_layer.0.weight", (vocab, 2 * H)))
    shapes.append(("prob_layer.0.bias", (vocab,)))
    return shapes


def closed_form_state(hp, vocab, lstm_scale=0.3, prob_scale=1.0, bridge_scale=0.2, blank_bias=1.0):
    """Deterministic state dict (numpy float32).  Conv weights/biases use the reference's own init range
    (uniform +-0.08, cnnlstm.py:158-159); BN gamma is centred on 1 so activations do not collapse, running
    stats are non-trivial (exercise eval mode); LSTM / bridge / prob ranges are widened so the recurrent
    dynamics are lively and greedy argmax margins are well separated (SURVEY.md §7 'Bit-exact greedy
    labels'); the blank logit gets a positive bias so blank frames occur."""
    sd = {}
    for name, shape in param_shapes(hp, vocab):
        if name.endswith("running_mean"):
            sd[name] = fill(shape, name, 0.05)
        elif name.endswith("running_var"):
            sd[name] = fill(shape, name, 0.2, offset=1.0)
        elif name.startswith("cnn.") and len(shape) == 1 and name.endswith("weight"):
            sd[name] = fill(shape, name, 0.3, offset=1.0)
        elif name.startswith("lstm."):
            sd[name] = fill(shape, name, lstm_scale)
        elif name.startswith("prob_layer"):
            sd[name] = fill(shape, name, prob_scale)
            if name.endswith("bias"):
                sd[name][0] += np.float32(blank_bias)
        elif name.startswith("bridge_layer"):
            sd[name] = fill(shape, name, bridge_scale)
        else:
            sd[name] = fill(shape, name, 0.08)
    return sd


def closed_form_batch(B, C, Himg, widths, vocab, labels_per_line, seed=1):
    """Synthetic batch in the SortByWidthCollater layout (reference: src/datautils.py:61-176):
    x zero-padded to the widest line, widths sorted descending, flat int32 targets in [1, vocab-1].
    Lines are blocky random strokes (5x6-pixel cells + fine noise) so features vary along the width."""
    r = np.random.RandomState(seed)
    Wmax = int(max(widths))
    x = np.zeros((B, C, Himg, Wmax), dtype=np.float32)
    for b in range(B):
        w = int(widths[b])
        for c in range(C):
            lo = r.uniform(0, 1, size=(Himg // 5 + 1, w // 6 + 1))
            img = np.repeat(np.repeat(lo, 5, 0), 6, 1)[:Himg, :w]
            x[b, c, :, :w] = (0.9 * img + 0.1 * r.uniform(0, 1, size=(Himg, w))).astype(np.float32)
    tl = np.asarray(labels_per_line, dtype=np.int32)
    t = r.randint(1, vocab, size=int(tl.sum())).astype(np.int32)
    return x, np.asarray(widths, dtype=np.int32), t, tl


def closed_form_pool_samples(B, seed=5):
    """Explicit FractionalMaxPool2d samples u in [0,1): shapes (B,64,2) and (B,128,2)."""
    r = np.random.RandomState(seed)
    s1 = r.uniform(0, 1, size=(B, 64, 2)).astype(np.float32)
    s2 = r.uniform(0, 1, size=(B, 128, 2)).astype(np.float32)
    return np.minimum(s1, np.float32(0.99999)), np.minimum(s2, np.float32(0.99999))
