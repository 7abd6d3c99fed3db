"""Greedy (best-path) CTC decoding: ArgmaxDecoder (src/decoder.py:112-185) == CnnOcrModel.decode_without_lm
(src/models/cnnlstm.py:479-541).  The per-frame argmax/max runs on the GPU (vocr_argmax_rows); one D2H copy of
int32[T,B] + float32[T,B] replaces the reference's T per-frame copies of the full logits."""
import numpy as np
import torch

from . import ops
from .textutils import uxxxx_to_utf8


def _frame_argmax(model_output):
    idx, mx = ops.argmax_rows(model_output.detach())
    return idx.cpu().numpy(), mx.cpu().numpy()


def greedy_label_sequences(model_output, batch_actual_timesteps, alphabet):
    """Integer label sequences (the emitted argmax indices) and uxxxx strings, with the reference's rules:
    blank resets; a maximum below 3/len(alphabet) (RAW activation) resets; repeats collapse by comparing the
    alphabet strings."""
    min_prob_thresh = 3 * 1 / len(alphabet)
    T, B = model_output.size()[0], model_output.size()[1]
    argmax_idxs, argmaxs = _frame_argmax(model_output)
    lens = [int(v) for v in batch_actual_timesteps]
    labels = [[] for _ in range(B)]
    result = ["" for _ in range(B)]
    idx_to_char = alphabet.idx_to_char
    for b in range(B):
        n = min(lens[b], T)
        ks = argmax_idxs[:n, b]
        low = argmaxs[:n, b] < min_prob_thresh          # same numpy float32-vs-python-float comparison as the reference
        prev = ""
        toks = []
        for t in range(n):
            k = int(ks[t])
            if k == 0 or low[t]:
                prev = ""
                continue
            ch = idx_to_char[k]
            if prev == ch:
                continue
            toks.append(ch)
            labels[b].append(k)
            prev = ch
        result[b] = " ".join(toks)
    return result, labels


def decode_greedy(model_output, batch_actual_timesteps, alphabet, uxxxx=False):
    result, _ = greedy_label_sequences(model_output, batch_actual_timesteps, alphabet)
    if uxxxx == False:  # noqa: E712  (mirrors the reference's comparison)
        result = [uxxxx_to_utf8(r) for r in result]
    return result


def greedy_labels_device(model_output, batch_actual_timesteps, alphabet):
    """Device-side collapse (vocr_greedy_collapse): returns a list of int lists without any host loop over T.
    The threshold is applied in float32 on the device."""
    dev = model_output.device
    idx, mx = ops.argmax_rows(model_output.detach())
    lens_dev = torch.as_tensor([int(v) for v in batch_actual_timesteps], dtype=torch.int32).to(dev)
    canon = torch.as_tensor(alphabet.canonical_indices(), dtype=torch.int32).to(dev)
    labels, counts = ops.greedy_collapse(idx, mx, lens_dev, canon, np.float32(3 * 1 / len(alphabet)))
    labels, counts = labels.cpu().numpy(), counts.cpu().numpy()
    return [[int(v) for v in labels[b, :counts[b]]] for b in range(labels.shape[0])]


class ArgmaxDecoder:
    def __init__(self, alphabet):
        self.alphabet = alphabet

    def decode(self, model_output, batch_actual_timesteps, uxxxx=False, lang=None):
        alphabet = self.alphabet if lang is None else self.alphabet[lang]
        return decode_greedy(model_output, batch_actual_timesteps, alphabet, uxxxx=uxxxx)
