"""Line-image dataset reader and host-side image transforms.

Restates the behaviour of the reference's `OcrDataset` (src/ocr_dataset.py:16-209) and of the three transforms the
reference's drivers compose for this path, `Scale(new_h=...)` -> `InvertBlackWhite()` -> `ToTensor()`
(src/imagetransforms.py:383-385,423-434,453-492; composed at src/decode_testset.py:48-65 and
src/train_cnn_lstm.py:215-243).  Host code only: it produces the `(image[C,H,W] float, label indices, metadata)`
items that `loop.SortByWidthCollater` batches for `CnnOcrModel`.

What is NOT pinned here: the reference decodes the LMDB payload with `cv2.imdecode` and rescales with `cv2.resize`; neither
`cv2` nor `lmdb` exists in the build image, so
  * `LmdbImageStore` imports both lazily and fails loudly without them;
  * `NpyDirImageStore` (one decoded `<id>.npy` array per line) is the store the tests use;
  * `Scale` calls `cv2.resize` when OpenCV is importable.  The reference passes its `interpolation=cv2.INTER_CUBIC` POSITIONALLY
    (src/imagetransforms.py:492), i.e. into cv2.resize's third parameter `dst`, so OpenCV resamples with its default,
    INTER_LINEAR - bilinear, not cubic.  Without OpenCV, uint8 images go through `_resize_linear_u8_cv`, a restatement of OpenCV's
    published uint8 bilinear path (11-bit fixed-point tap weights, INTER_RESIZE_COEF_BITS; 2x2 box average for an exact halving);
    it is pinned by hand-computed known answers and stays within 1 grey level of the float64 bilinear (tests/test_dataset_cpu.py),
    but could not be compared with a cv2 binary here: "parity unpinned" against OpenCV itself.  Images already at the model's
    line height never touch that code.
"""
import json
import os

import numpy as np
import torch

from .alphabet import Alphabet

SIZE_GROUP_LIMITS = [150, 200, 300, 350, 450, 600, np.inf]      # src/ocr_dataset.py:59


# ----------------------------------------------------------------------------------------------- image stores
class NpyDirImageStore:
    """`<dir>/<id>.npy`: the decoded pixels of one line (uint8, H x W or H x W x C)."""

    def __init__(self, directory):
        self.directory = directory

    def get(self, utt_id):
        return np.load(os.path.join(self.directory, utt_id + ".npy"))


class LmdbImageStore:
    """`line-images.lmdb`: key = utterance id (ASCII) -> encoded image bytes (src/ocr_dataset.py:42-46,156-157)."""

    def __init__(self, path):
        try:
            import cv2          # noqa: F401
            import lmdb
        except ImportError as e:            # pragma: no cover - neither module exists in the build image
            raise ImportError("LmdbImageStore needs the `lmdb` and `cv2` modules (%s); use NpyDirImageStore or pass "
                              "your own object with a get(id) -> ndarray method" % e)
        self._env = lmdb.Environment(path, map_size=int(1e12), readonly=True, lock=False)
        self._txn = self._env.begin(buffers=True)

    def get(self, utt_id):                  # pragma: no cover
        import cv2
        return cv2.imdecode(np.asarray(self._txn.get(utt_id.encode("ascii")), dtype=np.uint8), -1)


# ------------------------------------------------------------------------------------------------- transforms
class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img


def _cubic_weights(frac, a=-0.75):
    """OpenCV's bicubic kernel taps for sample offsets -1, 0, 1, 2 around a source position with fraction `frac`."""
    x = np.stack([frac + 1.0, frac, 1.0 - frac, 2.0 - frac], axis=-1)
    w = np.where(x <= 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0, ((a * x - 5.0 * a) * x + 8.0 * a) * x - 4.0 * a)
    return w


def _resize_bicubic(img, new_w, new_h):
    src = img.astype(np.float64)
    h, w = src.shape[:2]

    def taps(n_out, n_in):
        pos = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
        base = np.floor(pos).astype(np.int64)
        idx = np.clip(base[:, None] + np.arange(-1, 3)[None, :], 0, n_in - 1)
        return idx, _cubic_weights(pos - base)

    iy, wy = taps(new_h, h)
    ix, wx = taps(new_w, w)
    rows = (src[iy] * wy.reshape(wy.shape + (1,) * (src.ndim - 1))).sum(axis=1)            # [new_h, w, ...]
    cols = np.moveaxis(rows, 1, 0)[ix]                                                      # [new_w, 4, new_h, ...]
    out = (cols * wx.reshape(wx.shape + (1,) * (cols.ndim - 2))).sum(axis=1)               # [new_w, new_h, ...]
    out = np.moveaxis(out, 0, 1)
    if img.dtype == np.uint8:
        out = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return out


class Scale:
    """src/imagetransforms.py:453-492, the `new_h` / `new_w` form: the other side follows the aspect ratio with the
    reference's truncation `int(w * float(new_h / h))`; a non-positive result falls back to width 1."""

    def __init__(self, new_h=None, new_w=None, preserve_aspect_ratio=True, interpolation="linear"):
        assert isinstance(new_h, int) or isinstance(new_w, int)
        self.new_h, self.new_w, self.preserve_aspect_ratio = new_h, new_w, preserve_aspect_ratio
        self.interpolation = interpolation          # only read by the no-OpenCV fallback: "linear" (default) or "cubic"

    def target_size(self, h, w):
        nh, nw = self.new_h or h, self.new_w or w
        if self.preserve_aspect_ratio and self.new_h is None:
            nh = int(h * float(self.new_w / w))
        if self.preserve_aspect_ratio and self.new_w is None:
            nw = int(w * float(self.new_h / h))
        if nw <= 0 or nh <= 0:
            nw = 1
        return nh, nw

    def __call__(self, img):
        h, w = img.shape[:2]
        nh, nw = self.target_size(h, w)
        if (nh, nw) == (h, w):
            return img
        try:
            import cv2
            # the reference's own call, argument for argument (src/imagetransforms.py:492): the interpolation flag is passed
            # POSITIONALLY, i.e. into cv2.resize's third parameter `dst`, so OpenCV resamples with its default, INTER_LINEAR
            return cv2.resize(img, (nw, nh), cv2.INTER_CUBIC)          # pragma: no cover - cv2 absent in the build image
        except ImportError:
            # without OpenCV: what the call above effectively does (INTER_LINEAR).  uint8: OpenCV's fixed-point arithmetic restated;
            # other dtypes: float64 bilinear with OpenCV's half-pixel sample positions
            if self.interpolation != "linear":
                return _resize_bicubic(img, nw, max(nh, 1))
            if img.dtype == np.uint8:
                return _resize_linear_u8_cv(img, nw, max(nh, 1))
            return _resize_bilinear(img, nw, max(nh, 1))


def _resize_linear_u8_cv(img, new_w, new_h):
    """cv2.resize(img, (new_w, new_h)) for uint8 with the default INTER_LINEAR, restated from OpenCV's resize.cpp (3.x / 4.x, the plain
    C++ path):
      * an exact halving in both directions is rerouted to INTER_AREA, whose 2x2 case is (a + b + c + d + 2) >> 2;
      * otherwise, per destination column dx: fx = float((dx + 0.5) * (src_w / new_w) - 0.5), sx = floor(fx), fx -= sx; sx < 0 ->
        (sx, fx) = (0, 0); sx >= src_w - 1 -> (sx, fx) = (src_w - 1, 0); tap weights cvRound((1 - fx) * 2048) and cvRound(fx * 2048)
        as int16 (INTER_RESIZE_COEF_BITS = 11; cvRound rounds half to even); rows likewise;
      * horizontal pass in int32: D = S[sx] * a0 + S[sx + 1] * a1; vertical pass and cast:
        dst = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2."""
    src = np.ascontiguousarray(img)
    h, w = src.shape[:2]
    if h == 2 * new_h and w == 2 * new_w:
        v = src.astype(np.int32)
        return ((v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2] + 2) >> 2).astype(np.uint8)

    def taps(n_out, n_in):
        scale = float(n_in) / float(n_out)
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s0 = np.floor(f).astype(np.int64)
        f = (f - s0.astype(np.float32)).astype(np.float32)
        lo, hi = s0 < 0, s0 >= n_in - 1
        s0 = np.where(lo, 0, np.where(hi, n_in - 1, s0))
        f = np.where(lo | hi, np.float32(0), f).astype(np.float32)
        w1 = np.rint(f * np.float32(2048)).astype(np.int64)              # np.rint = round half to even = cvRound
        w0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        return s0, np.minimum(s0 + 1, n_in - 1), w0, w1

    x0, x1, a0, a1 = taps(new_w, w)
    y0, y1, b0, b1 = taps(new_h, h)
    v = src.astype(np.int64)
    shape_a = (1, -1) + (1,) * (src.ndim - 2)
    rows = v[:, x0] * a0.reshape(shape_a) + v[:, x1] * a1.reshape(shape_a)          # [h, new_w, ...] scaled by 2^11
    shape_b = (-1, 1) + (1,) * (src.ndim - 2)
    out = (((b0.reshape(shape_b) * (rows[y0] >> 4)) >> 16) + ((b1.reshape(shape_b) * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def _resize_bilinear(img, new_w, new_h):
    src = img.astype(np.float64)
    h, w = src.shape[:2]

    def taps(n_out, n_in):
        pos = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
        base = np.floor(pos).astype(np.int64)
        frac = pos - base
        i0 = np.clip(base, 0, n_in - 1)
        i1 = np.clip(base + 1, 0, n_in - 1)
        return i0, i1, frac

    y0, y1, fy = taps(new_h, h)
    x0, x1, fx = taps(new_w, w)
    fy = fy.reshape((-1,) + (1,) * (src.ndim - 1))
    rows = src[y0] * (1.0 - fy) + src[y1] * fy                                            # [new_h, w, ...]
    fxs = fx.reshape((1, -1) + (1,) * (src.ndim - 2))
    out = rows[:, x0] * (1.0 - fxs) + rows[:, x1] * fxs
    if img.dtype == np.uint8:
        out = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return out


class ConvertGray:
    """src/imagetransforms.py:408-413 (cv2.COLOR_BGR2GRAY on 3-channel images, anything else passes through).  OpenCV's
    uint8 formula restated exactly: (B*1868 + G*9617 + R*4899 + 8192) >> 14  (= 0.114 B + 0.587 G + 0.299 R)."""

    def __call__(self, img):
        if img.ndim == 3 and img.shape[2] == 3:
            if img.dtype == np.uint8:
                v = img.astype(np.int64)
                return ((v[:, :, 0] * 1868 + v[:, :, 1] * 9617 + v[:, :, 2] * 4899 + 8192) >> 14).astype(np.uint8)
            return (0.114 * img[:, :, 0] + 0.587 * img[:, :, 1] + 0.299 * img[:, :, 2]).astype(img.dtype)
        return img


class ConvertColor:
    """src/imagetransforms.py:415-420 (cv2.COLOR_GRAY2BGR on anything that is not already 3-channel)."""

    def __call__(self, img):
        if img.ndim == 3 and img.shape[2] == 3:
            return img
        g = img if img.ndim == 2 else img[:, :, 0]
        return np.stack([g, g, g], axis=2)


class InvertBlackWhite:
    """src/imagetransforms.py:383-385: `-img + 255`, which is 255 - img also in uint8 wrap-around arithmetic."""

    def __call__(self, img):
        return -img + 255


class ToTensor:
    """src/imagetransforms.py:423-434: H x W (x C) in [0, 255] -> float tensor C x H x W in [0, 1]."""

    def __call__(self, pic):
        if pic.ndim == 2:
            h, w = pic.shape
            return torch.from_numpy(np.ascontiguousarray(pic)).view(1, h, w).float().div(255)
        return torch.from_numpy(np.ascontiguousarray(pic.transpose((2, 0, 1)))).float().div(255)


def decode_transforms(line_height, num_in_channels=1, cvt_gray=False):
    """The inference-time pipeline of src/decode_testset.py:48-65: [ConvertGray if --cvtGray] -> Scale(new_h = the model's
    line height) -> [InvertBlackWhite only for single-channel models] -> ToTensor."""
    steps = []
    if cvt_gray:
        steps.append(ConvertGray())
    steps.append(Scale(new_h=line_height))
    if num_in_channels == 1:
        steps.append(InvertBlackWhite())
    steps.append(ToTensor())
    return Compose(steps)


# ---------------------------------------------------------------------------------------------------- dataset
class OcrDataset(torch.utils.data.Dataset):
    """`desc.json` {train, validation, test: [{id, trans, width, height?, writer?}]} + an image store.

    Same public attributes as the reference (`alphabet`, `size_group_limits`, `size_group_keys`, `size_groups`,
    `size_groups_dict`, `writer_id_map`, `nentries`, `max_index`) so `loop.GroupedSampler` and the training shell
    work on it unchanged; `__getitem__` returns `(image, label indices, metadata)` exactly like the reference."""

    def __init__(self, data_dir, split, transforms, alphabet=None, image_store=None, max_allowed_width=1200):
        self.max_allowed_width = max_allowed_width
        self.data_dir, self.split, self.preprocess = data_dir, split, transforms
        with open(os.path.join(data_dir, "desc.json"), "r") as fh:
            self.data_desc = json.load(fh)
        self.alphabet = alphabet if alphabet is not None else self._infer_alphabet()
        if image_store is None:
            npy_dir = os.path.join(data_dir, "line-images")
            image_store = NpyDirImageStore(npy_dir) if os.path.isdir(npy_dir) else LmdbImageStore(
                os.path.join(data_dir, "line-images.lmdb"))
        self.images = image_store

        # width groups at a nominal 30-px line height (src/ocr_dataset.py:59-96): an entry joins the first group whose
        # limit exceeds its normalised width, provided it is under max_allowed_width; wider lines are dropped
        self.size_group_limits = list(SIZE_GROUP_LIMITS)
        self.size_group_keys = self.size_group_limits
        self.size_groups = {k: [] for k in self.size_group_limits}
        self.size_groups_dict = {k: dict() for k in self.size_group_limits}
        self.writer_id_map = dict()
        for idx, entry in enumerate(self.data_desc[self.split]):
            if "writer" in entry and entry["writer"] not in self.writer_id_map:
                self.writer_id_map[entry["writer"]] = len(self.writer_id_map)
            if "height" in entry and "width" in entry:
                normalized_width = entry["width"] * (30 / entry["height"])
            elif "width" in entry:
                normalized_width = entry["width"]
            else:
                raise Exception("Json entry must list width & height of image.")
            for limit in self.size_group_limits:
                if normalized_width < limit and normalized_width < self.max_allowed_width:
                    self.size_groups[limit].append(idx)
                    self.size_groups_dict[limit][idx] = 1
                    break
        self.nentries = sum(len(v) for v in self.size_groups.values())
        self.max_index = max([max(v) for v in self.size_groups.values() if v], default=0)

    def _infer_alphabet(self):
        """src/ocr_dataset.py:109-120: CTC blank first, then every token of every split's transcriptions, sorted."""
        unique = set()
        for split in ("train", "validation", "test"):
            for entry in self.data_desc.get(split, []):
                unique.update(entry["trans"].split())
        return Alphabet(["<ctc-blank>", *sorted(unique)])

    def __getitem__(self, index):
        entry = self.data_desc[self.split][index]
        img = self.images.get(entry["id"])
        if img.ndim == 3 and img.shape[2] == 4:          # RGBA: drop alpha (src/ocr_dataset.py:164-166)
            img = img[:, :, :3]
        line_image = self.preprocess(img)
        if line_image.size(2) < 15:                      # narrower than the CNN can take: pad to 15 px with ONES (:173-176)
            padded = torch.ones(line_image.size(0), line_image.size(1), 15)
            padded[:, :, :line_image.size(2)] = line_image
            line_image = padded
        metadata = {"utt-id": entry["id"], "width": line_image.size(2)}
        if "writer" in entry:
            metadata["writer-id"] = self.writer_id_map[entry["writer"]]
        transcription = [self.alphabet.char_to_idx[ch] for ch in entry["trans"].split()]
        return line_image, transcription, metadata

    def __len__(self):
        return self.nentries
