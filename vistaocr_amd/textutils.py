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
    """Words of a uxxxx character sequence as the reference's scorer forms them (behaviour of src/textutils.py:290-323): u0020
    separates words, every punctuation mark and digit is a word of its own, the characters of a word are joined by "_".  With
    `with_spaces` a u0020 token follows each word that a space ended."""
    words, letters = [], []

    def close_word():
        if letters:
            words.append("_".join(letters))
            del letters[:]
            return True
        return False

    for ch in chars:
        if ch == "u0020":
            if close_word() and with_spaces:
                words.append("u0020")
        elif ch in _PUNCT or ch in _DIGITS:
            close_word()
            words.append(ch)
        else:
            letters.append(ch)
    close_word()
    return words


def _without_outer_spaces(words):
    lo, hi = 0, len(words)
    while lo < hi and words[lo] == "u0020":
        lo += 1
    while hi > lo and words[hi - 1] == "u0020":
        hi -= 1
    return words[lo:hi]


def compute_cer_wer(hyp_transcription, ref_transcription):
    """(CER, WER) of a hypothesis against a reference, both in uxxxx form (behaviour of src/textutils.py:326-351): character and word
    edit distances, each over the reference's length."""
    hyp_chars, ref_chars = hyp_transcription.split(" "), ref_transcription.split(" ")
    hyp_words = _without_outer_spaces(form_tokenized_words(hyp_chars))
    ref_words = _without_outer_spaces(form_tokenized_words(ref_chars))
    return float(edit_distance(hyp_chars, ref_chars)) / len(ref_chars), float(edit_distance(hyp_words, ref_words)) / len(ref_words)


def form_target_transcription(target, alphabet):
    """src/textutils.py:354-355."""
    return " ".join([alphabet.idx_to_char[int(i)] for i in target])
