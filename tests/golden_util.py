"""Helpers shared by the oracle-vs-golden (CPU) and HIP-vs-golden (GPU) tests."""
import ast
import os

import numpy as np

from oracle import closed_form as cf

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=True)


def alphabet_chars(which):
    return [str(c) for c in load("alphabets")[which]]


def case_inputs(name, which_alphabet="english"):
    """Rebuild (hp, chars, state, x, widths, targets, target_lens, pool samples) of a golden case."""
    g = load(name)
    hp = dict(ast.literal_eval(str(g["hp_json"])))
    chars = alphabet_chars(which_alphabet)
    V = len(chars)
    sd = cf.closed_form_state(hp, V)
    for k in g.files:
        if k.startswith("state/"):
            sd[k[len("state/"):]] = g[k]
    B = int(g["B"])
    x, w, tgt, tl = cf.closed_form_batch(B, hp.get("num_in_channels", 1), hp["input_line_height"],
                                         [int(v) for v in g["widths"]], V, [int(v) for v in g["labels_per_line"]],
                                         seed=int(g["batch_seed"]))
    s1, s2 = cf.closed_form_pool_samples(B)
    return g, hp, chars, sd, x, w, tgt, tl, (s1, s2)


def split_labels(g):
    flat, lens = g["labels_flat"], g["labels_len"]
    out, o = [], 0
    for n in lens:
        out.append([int(v) for v in flat[o:o + int(n)]])
        o += int(n)
    return out
