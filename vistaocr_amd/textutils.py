"""Text helpers of the hot path's boundary: uxxxx <-> utf8 and the CER/WER metric that BASELINE.json's
"CER vs ref" needs.  Restates src/textutils.py:216-355 of the reference (the reference module itself cannot
be imported: it needs icu_bidi and private files at import time)."""
import numpy as np

_PUNCT = set(("u002e u002c u003b u0027 u0022 u002f u0021 u0028 u0029 u005b u005d u003c u003e u002d u005f u007b u007d "
              "u0024 u0025 u0023 u0026 u060c u201d u060d u060f u061f u066d ufd3e ufd3f u061e u066a u066b u066c u002a "
              "u002b u003a u003d u005e u0060 u007c u007e").split())
_DIGITS = set(["u%04x" % c for c in range(0x660, 0x66a)] + ["u%04x" % c for c in range(0x30, 0x3a)])


def uxxxx_to_utf8(in_str):
    """src/textutils.py:216-243 — 'u0061 u0062' -> 'ab'; <unk>, <s>, </s> pass through."""
    if in_str.strip() == "":
        return ""
    result = ""
    for tok in in_str.split():
        if tok == "":
            continue
        if tok in ("<unk>", "<s>", "</s>"):
            result += tok
        else:
            result += chr(int(tok[1:], 16))
    return result


def utf8_to_uxxxx(in_str, output_array=False):
    """src/textutils.py:245-255."""
    arr = ["u%s" % hex(ord(ch))[2:].zfill(4).lower() for ch in in_str]
    return arr if output_array else " ".join(arr)


def edit_distance(A, B):
    """src/textutils.py:264-287 (Levenshtein, unit costs); row-vectorised DP instead of the O(n*m) Python loop."""
    la, lb = len(A), len(B)
    if la == 0 and lb == 0:
        return 0
    if la == 0 or lb == 0:
        return la + lb
    prev = np.arange(lb + 1, dtype=np.float64)
    b_arr = list(B)
    for i in range(1, la + 1):
        cur = np.empty(lb + 1, dtype=np.float64)
        cur[0] = i
        ai = A[i - 1]
        neq = np.fromiter((0.0 if ai == bj else 1.0 for bj in b_arr), dtype=np.float64, count=lb)
        # substitution/copy and deletion are elementwise; insertion is a running minimum along the row
        base = np.minimum(prev[:-1] + neq, prev[1:] + 1.0)
        run = cur[0]
        for j in range(lb):
            run = min(base[j], run + 1.0)
            cur[j + 1] = run
        prev = cur
    return prev[-1]


def form_tokenized_words(chars, with_spaces=False):
    """src/textutils.py:290-323 — split on u0020, punctuation and digits become single-char words."""
    words = []
    start = 0
    for i in range(len(chars)):
        if chars[i] == "u0020":
            if start != i:
                words.append("_".join(chars[start:i]))
                if with_spaces:
                    words.append("u0020")
            start = i + 1
            continue
        if chars[i] in _PUNCT or chars[i] in _DIGITS:
            if start != i:
                words.append("_".join(chars[start:i]))
            words.append(chars[i])
            start = i + 1
            continue
        if i == len(chars) - 1:
            if start == i:
                words.append(chars[start])
            else:
                words.append("_".join(chars[start:]))
    return words


def compute_cer_wer(hyp_transcription, ref_transcription):
    """src/textutils.py:326-351 — inputs in uxxxx form; returns (CER, WER)."""
    hyp_chars = hyp_transcription.split(" ")
    ref_chars = ref_transcription.split(" ")
    char_dist = edit_distance(hyp_chars, ref_chars)
    hyp_words = form_tokenized_words(hyp_chars)
    ref_words = form_tokenized_words(ref_chars)
    while len(hyp_words) > 0 and hyp_words[0] == "u0020":
        hyp_words = hyp_words[1:]
    while len(hyp_words) > 0 and hyp_words[-1] == "u0020":
        hyp_words = hyp_words[:-1]
    while len(ref_words) > 0 and ref_words[0] == "u0020":
        ref_words = ref_words[1:]
    while len(ref_words) > 0 and ref_words[-1] == "u0020":
        ref_words = ref_words[:-1]
    word_dist = edit_distance(hyp_words, ref_words)
    return float(char_dist) / len(ref_chars), float(word_dist) / len(ref_words)


def form_target_transcription(target, alphabet):
    """src/textutils.py:354-355."""
    return " ".join([alphabet.idx_to_char[int(i)] for i in target])
