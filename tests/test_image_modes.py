"""Image modes on the host side (no GPU): which pixel container a decoded PIL image becomes — i.e. which of Pillow's mode-dependent resize
behaviours the device will reproduce for it (tests/test_preprocess_gpu.py checks the pixels) — and the planning half of mq_resize_mode_u8."""
import numpy as np
from PIL import Image

from marqo_amd import _lib as L
from marqo_amd.engine import preprocess as P
from marqo_amd.s2_inference.image_input import pil_to_pixels, pil_to_rgb_u8
from oracle import preprocess as OP


def _rgba(h, w, seed=0, opaque=False):
    a = np.random.default_rng(seed).integers(0, 256, (h, w, 4), dtype=np.uint8)
    if opaque:
        a[..., 3] = 255
    return a


def test_pixel_container_follows_the_image_mode():
    a = _rgba(20, 30)
    assert isinstance(pil_to_pixels(Image.fromarray(a, "RGBA")), P.Rgba)
    la = pil_to_pixels(Image.fromarray(np.ascontiguousarray(a[..., [1, 3]]), "LA"))
    assert isinstance(la, P.Rgba) and np.array_equal(la.array[..., 0], a[..., 1]) and np.array_equal(la.array[..., 2], a[..., 1])
    assert np.array_equal(la.array[..., 3], a[..., 3]) and la.shape == (20, 30, 3)
    # an opaque alpha band changes nothing in Pillow's premultiplied round trip: such images ride the batched RGB path
    op = pil_to_pixels(Image.fromarray(_rgba(20, 30, opaque=True), "RGBA"))
    assert isinstance(op, np.ndarray) and op.shape == (20, 30, 3)
    pal = Image.fromarray(a[..., 0].copy(), "P")
    pal.putpalette([int(v) for v in np.random.default_rng(1).integers(0, 256, 768)])
    n = pil_to_pixels(pal)
    assert isinstance(n, P.NearestRgb) and np.array_equal(n.array, np.asarray(pal.convert("RGB")))
    assert isinstance(pil_to_pixels(Image.fromarray(a[..., 0].copy()).convert("1")), P.NearestRgb)
    lum = pil_to_pixels(Image.fromarray(a[..., 0].copy(), "L"))
    assert isinstance(lum, np.ndarray) and lum.shape == (20, 30, 3) and np.array_equal(lum[..., 0], lum[..., 2])
    rgb = pil_to_pixels(Image.fromarray(np.ascontiguousarray(a[..., :3]), "RGB"))
    assert isinstance(rgb, (np.ndarray, P.Rgbx)) and rgb.shape == (20, 30, 3)
    assert [P._kind(x) for x in (la, n, lum)] == [L.MQ_IMG_RGBA, L.MQ_IMG_NEAREST, L.MQ_IMG_RGB]


def test_flatten_is_the_plain_convert():
    a = _rgba(9, 7)
    im = Image.fromarray(a, "RGBA")
    assert np.array_equal(P.flatten_pixels(P.pil_pixels(im)), np.asarray(im.convert("RGB")))
    assert np.array_equal(pil_to_rgb_u8(im), a[..., :3])


def test_why_the_mode_matters():
    """the reference's transform on a translucent image (Pillow, image in its own mode) is NOT flatten-then-resize, and a palette image is
    resized with NEAREST: the two behaviours the device path has to follow"""
    a = _rgba(300, 400, seed=3)
    im = Image.fromarray(a, "RGBA")
    ref = OP.clip_resize_crop_pil_image(im, 224)
    flat = OP.clip_resize_crop_u8(np.asarray(im.convert("RGB")), 224, backend="pil")
    assert ref.shape == flat.shape == (224, 224, 3) and not np.array_equal(ref, flat)
    pal = Image.fromarray(a[..., 0].copy(), "P")
    assert np.array_equal(np.asarray(pal.resize((50, 60), Image.BICUBIC)), np.asarray(pal.resize((50, 60), Image.NEAREST)))
    same = Image.fromarray(_rgba(224, 300, seed=4), "RGBA")      # shorter side already 224: torchvision returns the image untouched
    assert np.array_equal(OP.clip_resize_crop_pil_image(same, 224), np.asarray(same.convert("RGB"))[:, 38:262])


def test_mode_planning_without_a_gpu():
    lib = L.load()
    h = np.array([300, 224, 64], dtype=np.int32)
    w = np.array([400, 224, 900], dtype=np.int32)
    rgb = lib.mq_resize_mode_workspace_bytes(h.ctypes.data, w.ctypes.data, 3, 224, 224, 3, 1, L.MQ_IMG_RGB)
    assert rgb == lib.mq_clip_resize_workspace_bytes(h.ctypes.data, w.ctypes.data, 3, 224) > 0
    assert lib.mq_resize_mode_workspace_bytes(h.ctypes.data, w.ctypes.data, 3, 240, 200, 2, 0, L.MQ_IMG_RGB) == \
        lib.mq_resize_filter_workspace_bytes(h.ctypes.data, w.ctypes.data, 3, 240, 200, 2)
    assert lib.mq_resize_mode_workspace_bytes(h.ctypes.data, w.ctypes.data, 3, 224, 224, 3, 1, L.MQ_IMG_RGBA) > rgb      # four bands
    assert 0 < lib.mq_resize_mode_workspace_bytes(h.ctypes.data, w.ctypes.data, 3, 224, 224, 3, 1, L.MQ_IMG_NEAREST) < rgb  # one tap
    assert lib.mq_resize_mode_workspace_bytes(h.ctypes.data, w.ctypes.data, 3, 224, 200, 3, 1, L.MQ_IMG_RGB) == 0   # crop needs a square
    assert lib.mq_resize_mode_workspace_bytes(h.ctypes.data, w.ctypes.data, 3, 224, 224, 3, 1, 5) == 0              # unknown mode
    assert lib.mq_resize_mode_u8(None, None, None, None, 1, 224, 224, 3, 1, 0, None, None, 0, None) == -1
