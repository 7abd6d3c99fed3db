"""Dataset reader + host transforms (vistaocr_amd/dataset.py; reference: src/ocr_dataset.py, src/imagetransforms.py).
The reference's module cannot be imported here (cv2, lmdb absent), so these pin the restated rules on synthetic data."""
import json
import os

import numpy as np
import torch

from vistaocr_amd import dataset as ds
from vistaocr_amd.loop import GroupedSampler, SortByWidthCollater


def _make(tmp_path, entries_by_split, images):
    d = str(tmp_path)
    with open(os.path.join(d, "desc.json"), "w") as fh:
        json.dump(entries_by_split, fh)
    os.mkdir(os.path.join(d, "line-images"))
    for k, v in images.items():
        np.save(os.path.join(d, "line-images", k + ".npy"), v)
    return d


def test_grouping_alphabet_and_items(tmp_path):
    rng = np.random.RandomState(0)
    widths = [100, 149, 150, 299, 610, 1300, 40, 8]
    train = []
    images = {}
    for i, w in enumerate(widths):
        e = {"id": "utt%d" % i, "trans": "u0061 u0062 u0020 u0061" if i % 2 else "u0063", "width": w, "height": 30}
        if i < 3:
            e["writer"] = "w%d" % (i % 2)
        train.append(e)
        images[e["id"]] = rng.randint(0, 256, size=(30, w), dtype=np.uint8)
    # an entry listed at another height: its group follows the width normalised to 30 px (src/ocr_dataset.py:80-84)
    train.append({"id": "tall", "trans": "u0064", "width": 400, "height": 60})
    images["tall"] = rng.randint(0, 256, size=(60, 400), dtype=np.uint8)
    d = _make(tmp_path, {"train": train, "validation": [{"id": "v", "trans": "u007a", "width": 50, "height": 30}], "test": []}, images)
    data = ds.OcrDataset(d, "train", ds.decode_transforms(30))

    assert data.alphabet.char_array == ["<ctc-blank>", "u0020", "u0061", "u0062", "u0063", "u0064", "u007a"]
    g = data.size_groups
    assert g[150] == [0, 1, 6, 7] and g[200] == [2] and g[300] == [3, 8] and g[np.inf] == [4]
    assert all(5 not in v for v in g.values())                       # 1300 >= max_allowed_width: dropped
    assert len(data) == 8 and data.max_index == 8
    assert data.writer_id_map == {"w0": 0, "w1": 1}
    assert list(GroupedSampler(data, rand=False)) == [0, 1, 6, 7, 2, 3, 8, 4]

    img, labels, meta = data[1]
    assert img.shape == (1, 30, 149) and img.dtype == torch.float32
    expect = torch.from_numpy((255 - images["utt1"]).astype(np.float32) / 255).view(1, 30, 149)
    assert torch.equal(img, expect)                                   # no rescale at the native height: exact
    assert labels == [2, 3, 1, 2] and meta == {"utt-id": "utt1", "width": 149, "writer-id": 1}

    img, labels, meta = data[7]                                       # 8 px wide: padded to 15 with ones (src/ocr_dataset.py:173-176)
    assert img.shape == (1, 30, 15) and meta["width"] == 15
    assert torch.all(img[:, :, 8:] == 1.0)

    img, _, meta = data[8]                                            # 60 x 400 -> 30 x 200
    assert img.shape == (1, 30, 200) and meta["width"] == 200

    x, target, widths_t, target_lens, metas = SortByWidthCollater([data[i] for i in (7, 1, 0)])
    assert x.shape == (3, 1, 30, 149) and widths_t.tolist() == [149, 100, 15]
    assert target_lens.tolist() == [4, 1, 4] and target.tolist() == [2, 3, 1, 2, 4, 2, 3, 1, 2]
    assert metas["utt-ids"] == ["utt1", "utt0", "utt7"]


def test_scale_sizes_and_bicubic_properties():
    s = ds.Scale(new_h=30)
    assert s.target_size(60, 401) == (30, 200)                        # int(401 * 0.5): truncation, like the reference
    assert s.target_size(45, 2) == (30, 1)
    assert s.target_size(3000, 10) == (30, 1)                         # would be 0 wide: falls back to 1
    flat = np.full((60, 80), 77, dtype=np.uint8)
    assert np.all(s(flat) == 77)                                      # kernel taps sum to one
    ramp = np.tile(np.arange(0, 200, 2, dtype=np.uint8), (64, 1))     # linear in x: reproduced away from the border
    out = ds.Scale(new_h=32)(ramp).astype(np.int64)
    assert out.shape == (32, 50)
    assert np.all(np.abs(out[:, 2:-2] - (np.arange(50) * 4 + 1)[None, 2:-2]) <= 1)
    outc = ds.Scale(new_h=32, interpolation="cubic")(ramp).astype(np.int64)          # the bicubic fallback keeps a ramp too
    assert np.all(np.abs(outc[:, 2:-2] - (np.arange(50) * 4 + 1)[None, 2:-2]) <= 1)
    # exact bilinear value at a known position: 2x upsampling of [0, 100] samples 0.25 / 0.75 of the way
    up = ds.Scale(new_w=4, preserve_aspect_ratio=False)(np.array([[0, 100]], dtype=np.uint8))
    assert up.tolist() == [[0, 25, 75, 100]]
    rgba = np.zeros((30, 20, 4), dtype=np.uint8)
    assert ds.ToTensor()(rgba[:, :, :3]).shape == (3, 30, 20)
    assert np.array_equal(ds.InvertBlackWhite()(np.array([[0, 1, 255]], dtype=np.uint8)), np.array([[255, 254, 0]], dtype=np.uint8))


def test_fixed_point_bilinear_known_answers():
    """The no-OpenCV uint8 rescale restates OpenCV's INTER_LINEAR fixed-point path (11-bit weights; src/imagetransforms.py:492 ends up
    there because the reference passes its interpolation flag into `dst`).  Known answers worked out by hand from the published
    arithmetic, a case where it differs from float bilinear + round-half-even, the 2x2 box average of an exact halving, and the bound
    against the float64 bilinear: at most one grey level."""
    f = ds._resize_linear_u8_cv
    # [0, 100] -> 4 wide: taps 2048/0, 1536/512, 512/1536, 0/2048 -> D = 0, 51200, 153600, 204800 -> ((2048 * (D >> 4)) >> 16) + 2 >> 2
    assert f(np.array([[0, 100]], dtype=np.uint8), 4, 1).tolist() == [[0, 25, 75, 100]]
    # [10, 20, 40] -> 2 wide: f = 0.25 -> D = 10*1536 + 20*512 = 25600 -> (50 + 2) >> 2 = 13 (float bilinear: 12.5 -> 12); f = 0.75 -> 35
    assert f(np.array([[10, 20, 40]], dtype=np.uint8), 2, 1).tolist() == [[13, 35]]
    assert ds._resize_bilinear(np.array([[10, 20, 40]], dtype=np.uint8), 2, 1).tolist() == [[12, 35]]
    # exact halving in both directions -> INTER_AREA's 2x2 mean with rounding: (1 + 2 + 3 + 5 + 2) >> 2
    assert f(np.array([[1, 2], [3, 5]], dtype=np.uint8), 1, 1).tolist() == [[3]]
    r = np.random.RandomState(0)
    for (h, w, nh, nw) in ((45, 133, 30, 88), (64, 301, 30, 141), (22, 50, 30, 68), (60, 200, 30, 100), (31, 77, 30, 74)):
        img = r.randint(0, 256, size=(h, w)).astype(np.uint8)
        a = f(img, nw, nh).astype(np.int64)
        if (h, w) == (2 * nh, 2 * nw):
            ref = img.astype(np.float64).reshape(nh, 2, nw, 2).mean(axis=(1, 3))
        else:
            ref = ds._resize_bilinear(img.astype(np.float64), nw, nh)
        assert a.shape == (nh, nw) and float(np.abs(a - ref).max()) <= 1.0, (h, w, nh, nw)
    rgb = r.randint(0, 256, size=(40, 90, 3)).astype(np.uint8)
    out = f(rgb, 67, 30)
    assert out.shape == (30, 67, 3)
    for c in range(3):
        assert np.array_equal(out[:, :, c], f(rgb[:, :, c], 67, 30))              # channels are independent
    assert np.array_equal(ds.Scale(new_h=30)(rgb), f(rgb, 67, 30))                  # int(90 * 30 / 40) = 67: the Scale transform uses it


def test_decode_pipeline_follows_the_models_channel_count():
    """src/decode_testset.py:48-65: InvertBlackWhite only for single-channel models; ConvertGray only with --cvtGray;
    grey conversion = OpenCV's integer BGR2GRAY formula."""
    bgr = np.zeros((30, 40, 3), dtype=np.uint8)
    bgr[:, :, 0], bgr[:, :, 1], bgr[:, :, 2] = 10, 200, 90
    t3 = ds.decode_transforms(30, num_in_channels=3)(bgr)
    assert t3.shape == (3, 30, 40) and abs(float(t3[1, 0, 0]) - 200 / 255) < 1e-6          # colour model: not inverted
    t1 = ds.decode_transforms(30, num_in_channels=1, cvt_gray=True)(bgr)
    grey = (10 * 1868 + 200 * 9617 + 90 * 4899 + 8192) >> 14
    assert grey == 145 and t1.shape == (1, 30, 40) and abs(float(t1[0, 0, 0]) - (255 - grey) / 255) < 1e-6
    assert ds.ConvertColor()(np.full((5, 6), 9, dtype=np.uint8)).shape == (5, 6, 3)
    assert ds.ConvertGray()(np.full((5, 6), 9, dtype=np.uint8)).shape == (5, 6)


def test_lmdb_store_fails_loudly_without_its_modules(tmp_path):
    try:
        import cv2, lmdb        # noqa: F401,E401
    except ImportError:
        import pytest
        with pytest.raises(ImportError):
            ds.LmdbImageStore(str(tmp_path))
