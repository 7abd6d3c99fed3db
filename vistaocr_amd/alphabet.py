"""Alphabet surface of the reference (src/alphabet.py:1-12) plus the static symbol tables it ships
(src/english.py:16-33, src/arabic.py:23-53, src/french.py:26).  Index 0 is always '<ctc-blank>'.

The tables are data: they are stored as runs of hexadecimal code points and expanded to the reference's
'uXXXX' spelling; tests/test_host_cpu.py::test_alphabets_match_reference_tables checks them entry by entry against
tests/golden/alphabets.npz."""

BLANK = "<ctc-blank>"


class Alphabet(object):
    def __init__(self, char_array, left_to_right=False):
        self.left_to_right = left_to_right
        self.char_to_idx = dict(zip(char_array, range(len(char_array))))      # last duplicate wins
        self.idx_to_char = dict(zip(range(len(char_array)), char_array))
        self.char_array = char_array

    def __len__(self):
        return len(self.idx_to_char)

    def canonical_indices(self):
        """canon[k] = smallest index whose symbol STRING equals that of k: the reference's repeat collapse
        compares strings (cnnlstm.py:520-523), so duplicated entries (English 'u002d' at 73 and 91) merge."""
        first = {}
        out = []
        for k in range(len(self.idx_to_char)):
            ch = self.idx_to_char[k]
            first.setdefault(ch, k)
            out.append(first[ch])
        return out


def _expand(spec):
    out = [BLANK]
    for part in spec.split():
        if "-" in part:
            a, b = part.split("-")
            out.extend("u%04x" % c for c in range(int(a, 16), int(b, 16) + 1))
        else:
            out.append("u%04x" % int(part, 16))
    return out


_ENGLISH = ("61-7a 41-5a 30-39 20 2e 2c 3b 5b 5d 7b 7d 28 29 2d 3a 3c 3e 3f 2f 3d 2a 26 5e 25 24 23 40 21 7e 60 2b 2d "
            "5f 7c 22 27")
_ARABIC = ("20-22 24-58 5a-5f 61-7b 7d 7e a9 ab ad b7 bb be d7 e8 ec 60c 61b 61f 621-63a 640-652 66a 6d2 200c-200f "
           "2013 2014 2018 2019 201c 201d 2022 2026 25cf fe87 fef9")
_FRENCH = "20-22 25 27-3b 3d 3f 41-5a 5f 61-7b 7d a4 b0 b2 c0 c9 e0 e2 e7-eb ee f4 f9 fb 153 20ac"


def english_alphabet():
    return Alphabet(_expand(_ENGLISH), left_to_right=True)


def arabic_alphabet():
    return Alphabet(_expand(_ARABIC), left_to_right=False)


def french_alphabet():
    return Alphabet(_expand(_FRENCH), left_to_right=True)
