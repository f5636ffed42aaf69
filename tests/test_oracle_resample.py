"""Pins oracle/resample.c (restatement of Pillow's ImagingResample) bit-for-bit against Pillow itself and
against the committed golden fixture; also checks the product's host-side coefficient tables
(mq_resample_coeffs — host code, no GPU needed) against the oracle's."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import preprocess as OP

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resample_pil.npz")


def test_c_restatement_equals_pillow_random_sizes():
    rng = np.random.default_rng(0)
    for _ in range(40):
        h, w = int(rng.integers(1, 700)), int(rng.integers(1, 700))
        oh, ow = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(OP.resize_u8(img, ow, oh, "c"), OP.resize_u8(img, ow, oh, "pil")), (h, w, oh, ow)


def test_c_restatement_equals_pillow_extremes():
    rng = np.random.default_rng(1)
    for (h, w, oh, ow) in [(1, 1, 224, 224), (2, 3000, 224, 224), (3000, 31, 10, 10), (224, 224, 224, 224),
                           (224, 300, 224, 300), (300, 224, 224, 224), (80, 80, 224, 224), (4000, 3000, 240, 240)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(OP.resize_u8(img, ow, oh, "c"), OP.resize_u8(img, ow, oh, "pil")), (h, w, oh, ow)
    sat = np.zeros((50, 50, 3), np.uint8); sat[::2] = 255  # ringing must clip, not wrap
    assert np.array_equal(OP.resize_u8(sat, 123, 77, "c"), OP.resize_u8(sat, 123, 77, "pil"))


def test_tall_thin_special_case_of_newer_pillow():
    """Pillow >= 11 (12.2.0 is installed here) resizes images with height > 100 x width vertically FIRST
    (Image.resize in PIL/Image.py); Pillow 10.4.0 — the version the reference pins (requirements.dev.txt:29) —
    always runs the horizontal pass first, which is what the oracle and the GPU kernels implement.  Pin the
    understanding of that divergence so a fixture regenerated on a newer Pillow is not mis-read as a bug."""
    img = np.random.default_rng(7).integers(0, 256, (3000, 2, 3), dtype=np.uint8)
    pil = OP.resize_u8(img, 10, 10, "pil")
    v_then_h = OP.resize_u8(OP.resize_u8(img, 2, 10, "c"), 10, 10, "c")
    h_then_v = OP.resize_u8(img, 10, 10, "c")
    import PIL
    if int(PIL.__version__.split(".")[0]) >= 11:
        assert np.array_equal(pil, v_then_h)
    else:
        assert np.array_equal(pil, h_then_v)


def test_golden_fixture():
    z = np.load(GOLD)
    for i in range(int(z["n"])):
        img, out = z[f"img{i}"], z[f"clip{i}"]
        assert np.array_equal(OP.clip_resize_crop_u8(img, 224, "c"), out)
    patches, boxes = OP.chunk_image_simple(z["img0"], 3, 3, False, "c")
    assert len(patches) == 10
    for k, p in enumerate(patches):
        assert np.array_equal(p, z[f"chunk{k}"])
    assert np.allclose(np.asarray(boxes), z["chunk_boxes"])


def test_transform_helpers_follow_torchvision_rules():
    assert OP.resize_output_size(480, 640, 224) == (224, 298)
    assert OP.resize_output_size(640, 480, 224) == (298, 224)
    assert OP.resize_output_size(224, 500, 224) == (224, 500)
    assert OP.center_crop_offsets(224, 225, 224) == (0, 0)   # round(0.5) -> 0 (banker's)
    assert OP.center_crop_offsets(224, 227, 224) == (0, 2)   # round(1.5) -> 2
    assert OP.center_crop_offsets(298, 224, 224) == (37, 0)
    x = OP.clip_transform(np.full((300, 260, 3), 128, np.uint8))
    assert x.shape == (3, 224, 224) and x.dtype == np.float32
    assert abs(x[0, 0, 0] - (128 / 255 - OP.OPENAI_DATASET_MEAN[0]) / OP.OPENAI_DATASET_STD[0]) < 1e-6


def test_generate_boxes_known_answers():
    """reference tests/processing/test_image_utils.py:199-254 style known answers."""
    b = OP.generate_boxes((240, 240), 3, 3)
    assert len(b) == 9 and b[0] == (0, 0, 80, 80) and b[-1] == (160, 160, 240, 240)
    assert len(OP.generate_boxes((240, 240), 3, 3, overlap=True)) == 9 + 4
    assert len(OP.generate_boxes((240, 240), 7, 7)) == 49  # 240 // 7 = 34; 7 * 34 = 238 <= 240
    assert OP.rescale_box((0, 0, 240, 240), (240, 240), (640, 480)) == [0.0, 0.0, 640.0, 480.0]


def test_product_host_coefficients_match_oracle():
    from marqo_amd import _lib as L
    lib = L.load()
    for (n_in, n_out) in [(640, 224), (80, 224), (224, 298), (3000, 240), (7, 240), (225, 224)]:
        ob, ok = OP.coeffs(n_in, n_out)
        ks = lib.mq_resample_ksize(n_in, n_out)
        assert ks == ok.shape[1]
        first, count = n_out // 3, n_out - n_out // 3
        b = np.zeros((count, 2), np.int32); k = np.zeros((count, ks), np.int32)
        L.check(lib.mq_resample_coeffs(n_in, n_out, first, count, b.ctypes.data, k.ctypes.data))
        assert np.array_equal(b, ob[first:]) and np.array_equal(k, ok[first:])


def test_bilinear_filter_matches_pillow():
    """CLIPA's preprocessing (open_clip _apcfg: BILINEAR squash): the C restatement with the triangle filter == Pillow"""
    from PIL import Image
    rng = np.random.default_rng(31)
    for (h, w), (oh, ow) in [((224, 224), (224, 224)), ((300, 200), (224, 224)), ((17, 23), (224, 224)), ((1201, 1600), (224, 224)),
                              ((64, 500), (336, 336)), ((500, 64), (96, 160))]:
        im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(im).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(OP.resize_u8(im, ow, oh, backend="c", filt=OP.FILTER_BILINEAR), ref), (h, w, oh, ow)
