"""ORACLE — test infrastructure only.  Never imported by the product (marqo_amd/*).

CPU restatement of the image side of the vectorise() path that runs BEFORE the towers:

* ``clip_transform``  <- the transform returned by ``_get_transform`` (src/marqo/s2_inference/clip_utils.py:48-67)
                         / open_clip 2.24.0 ``image_transform_v2`` (open_clip_model.py:84-85):
                         Resize(n_px, BICUBIC) -> CenterCrop(n_px) -> RGB -> ToTensor -> Normalize.
                         torchvision 0.13.1 is not installable here; its Resize/CenterCrop on a PIL image are
                         restated from the published functional_pil semantics (shorter side -> n_px,
                         long side = int(n_px * long / short); crop offsets int(round((dim - n_px) / 2.0))).
* ``chunk_image_simple`` <- ``PatchifySimple`` + ``generate_boxes`` + ``rescale_box`` + ``patchify_image``
                         (src/marqo/s2_inference/processing/image.py:120-151,
                          src/marqo/s2_inference/processing/image_utils.py:141-202,267-279)

Two implementations of the resampler sit here side by side:
  ``backend="pil"``  calls Pillow itself (the third-party dependency the reference runs: Pillow==10.4.0
                     pinned, 12.2.0 installed here — same ImagingResample algorithm);
  ``backend="c"``    calls oracle/resample.c (our restatement of that algorithm, built by oracle/Makefile).
tests/test_oracle_resample.py pins "c" == "pil" bit-for-bit; the GPU kernels are then checked against "c"
(which travels to the GPU box, and equals PIL there too since Pillow is in the image).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Sequence, Tuple

import numpy as np

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)  # clip_utils.py:32
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)  # clip_utils.py:33

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

FILTER_BILINEAR, FILTER_BICUBIC = 2, 3


def _c():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "resample.c")):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        lib = C.CDLL(_LIB_PATH)
        lib.orc_resize_u8.restype = C.c_int
        lib.orc_resize_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.orc_precompute_coeffs.restype = C.c_int
        lib.orc_precompute_coeffs.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_int)),
                                              C.POINTER(C.POINTER(C.c_int32))]
        lib.orc_free.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def coeffs(in_size: int, out_size: int, filt: int = FILTER_BICUBIC) -> Tuple[np.ndarray, np.ndarray]:
    """(bounds int32 [out,2], kk int32 [out,ksize]) of one axis, from the C restatement."""
    lib = _c()
    b, k = C.POINTER(C.c_int)(), C.POINTER(C.c_int32)()
    ks = lib.orc_precompute_coeffs(in_size, 0.0, float(in_size), out_size, filt, C.byref(b), C.byref(k))
    bounds = np.ctypeslib.as_array(b, shape=(out_size, 2)).copy()
    kk = np.ctypeslib.as_array(k, shape=(out_size, ks)).copy()
    lib.orc_free(b); lib.orc_free(k)
    return bounds.astype(np.int32), kk.astype(np.int32)


def resize_u8(img: np.ndarray, out_w: int, out_h: int, backend: str = "c", filt: int = FILTER_BICUBIC) -> np.ndarray:
    """PIL.Image.resize((out_w, out_h), BICUBIC) of a uint8 [H, W, C] (or [H, W]) array."""
    assert img.dtype == np.uint8
    if backend == "pil":
        from PIL import Image
        res = {FILTER_BICUBIC: Image.BICUBIC, FILTER_BILINEAR: Image.BILINEAR}[filt]
        return np.asarray(Image.fromarray(img).resize((out_w, out_h), res))
    squeeze = img.ndim == 2
    a = np.ascontiguousarray(img[:, :, None] if squeeze else img)
    h, w, ch = a.shape
    out = np.empty((out_h, out_w, ch), dtype=np.uint8)
    rc = _c().orc_resize_u8(a.ctypes.data, h, w, ch, w * ch, out_h, out_w, filt, out.ctypes.data)
    assert rc == 0
    return out[:, :, 0] if squeeze else out


def resize_output_size(h: int, w: int, n_px: int) -> Tuple[int, int]:
    """torchvision 0.13 Resize(int): shorter side -> n_px, the other int(n_px * long / short).  Returns (new_h, new_w)."""
    short, long = (w, h) if w <= h else (h, w)
    if short == n_px:
        return h, w
    new_short, new_long = n_px, int(n_px * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def center_crop_offsets(h: int, w: int, n_px: int) -> Tuple[int, int]:
    """torchvision CenterCrop: top = int(round((h - n_px) / 2.0)) (Python banker's rounding), same for left."""
    return int(round((h - n_px) / 2.0)), int(round((w - n_px) / 2.0))


def clip_resize_crop_u8(img: np.ndarray, n_px: int = 224, backend: str = "c") -> np.ndarray:
    """uint8 RGB [H, W, 3] -> uint8 [n_px, n_px, 3]: the integer part of the CLIP transform.
    (Images smaller than n_px after the resize cannot happen: the shorter side becomes n_px.)"""
    h, w = img.shape[:2]
    nh, nw = resize_output_size(h, w, n_px)
    r = resize_u8(img, nw, nh, backend=backend)
    top, left = center_crop_offsets(nh, nw, n_px)
    return np.ascontiguousarray(r[top:top + n_px, left:left + n_px])


def clip_resize_crop_pil_image(img, n_px: int = 224) -> np.ndarray:
    """The integer part of the reference's CLIP transform applied to a PIL image IN ITS OWN MODE, with Pillow itself
    (src/marqo/s2_inference/clip_utils.py:61-64: Resize(n_px, BICUBIC) -> CenterCrop(n_px) -> _convert_image_to_rgb; torchvision
    0.13 functional_pil.resize: an image whose shorter side already equals n_px is returned untouched, otherwise
    img.resize((new_w, new_h), BICUBIC)).  Image.resize is mode dependent — NEAREST for "P" / "1", a premultiplied round trip for
    "RGBA" / "LA" — which is exactly what this checker exists for.  -> uint8 [n_px, n_px, 3]."""
    from PIL import Image
    w, h = img.size
    nh, nw = resize_output_size(h, w, n_px)
    r = img if (nh, nw) == (h, w) else img.resize((nw, nh), Image.BICUBIC)
    top, left = center_crop_offsets(nh, nw, n_px)
    r = r.crop((left, top, left + n_px, top + n_px))
    return np.ascontiguousarray(np.asarray(r.convert("RGB")))


def squash_pil_image(img, out_h: int, out_w: int, bilinear: bool = False) -> np.ndarray:
    """Image.resize((out_w, out_h), BICUBIC | BILINEAR) of a PIL image in its own mode, then convert("RGB") (the SigLIP / CLIPA
    squash transforms and the chunker's working image, processing/image.py:143)."""
    from PIL import Image
    return np.ascontiguousarray(np.asarray(img.resize((out_w, out_h), Image.BILINEAR if bilinear else Image.BICUBIC).convert("RGB")))


def to_tensor_normalize(u8: np.ndarray, mean: Sequence[float] = OPENAI_DATASET_MEAN, std: Sequence[float] = OPENAI_DATASET_STD) -> np.ndarray:
    """ToTensor (/255, HWC -> CHW, fp32) + Normalize."""
    x = u8.astype(np.float32) / np.float32(255.0)
    x = (x - np.asarray(mean, dtype=np.float32)) / np.asarray(std, dtype=np.float32)
    return np.ascontiguousarray(np.moveaxis(x, -1, -3))


def clip_transform(img: np.ndarray, n_px: int = 224, mean=OPENAI_DATASET_MEAN, std=OPENAI_DATASET_STD, backend: str = "c") -> np.ndarray:
    return to_tensor_normalize(clip_resize_crop_u8(img, n_px, backend), mean, std)


# ---- simple / overlap grid chunker ------------------------------------------------------------------
CHUNK_SIZE = (240, 240)  # image_utils.py:16-22


def generate_boxes(image_size: Tuple[int, int], hn: int, wn: int, overlap: bool = False) -> List[Tuple[int, int, int, int]]:
    """image_utils.py:165-202 (the `continue` inside the overlap branch also skips nothing further: it is the last
    statement of the loop body)."""
    img_width, img_height = image_size
    height, width = img_height // hn, img_width // wn
    boxes = []
    for i in range(0, img_height, height):
        for j in range(0, img_width, width):
            p1, p2 = j + width, i + height
            if p1 > img_width or p2 > img_height:
                continue
            boxes.append((j, i, p1, p2))
            if overlap:
                p3, p4 = p1 + width // 2, p2 + height // 2
                if p3 > img_width or p4 > img_height:
                    continue
                boxes.append((j + width // 2, i + height // 2, p3, p4))
    return boxes


def rescale_box(box, from_size, to_size) -> List[float]:
    fy, fx = to_size[1] / from_size[1], to_size[0] / from_size[0]
    x1, y1, x2, y2 = box
    return [x1 * fx, y1 * fy, x2 * fx, y2 * fy]


def chunk_image_simple(img: np.ndarray, hn: int = 3, wn: int = 3, overlap: bool = False, backend: str = "c"):
    """-> (patches: list of uint8 arrays, first = the whole 240x240 working image; bboxes in original pixel coords)."""
    h, w = img.shape[:2]
    resized = resize_u8(img, CHUNK_SIZE[0], CHUNK_SIZE[1], backend=backend)
    boxes = [(0, 0, CHUNK_SIZE[0], CHUNK_SIZE[1])] + generate_boxes(CHUNK_SIZE, hn, wn, overlap)
    patches = [np.ascontiguousarray(resized[y1:y2, x1:x2]) for (x1, y1, x2, y2) in boxes]
    return patches, [rescale_box(bb, CHUNK_SIZE, (w, h)) for bb in boxes]
