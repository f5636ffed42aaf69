"""Image chunking ('simple' / 'overlap' patch methods) — the model-free chunkers of the reference
(src/marqo/s2_inference/processing/image.py:46-151, image_utils.py:16-22,141-202,267-307).

`chunk_image(image, device, method)` keeps the reference's signature and return value
`(patches: List[PIL.Image], bboxes_orig: List[[x1, y1, x2, y2] floats])` — the whole 240x240 working image first —
but the resampling (the only arithmetic in it) runs on the GPU, bit-identical to Pillow.  The model-based chunkers
(frcnn / yolox / dino) are separate detector models and out of scope.

`chunk_images_to_tensors` is the engine's fused form: for a CLIP model it returns the crops already resized /
centre-cropped / normalised on the device (K11), ready for `encode_image` without a host round trip.
"""
from __future__ import annotations

import threading
from typing import List, Tuple, Union
from urllib.parse import urlparse

import numpy as np
import PIL
from PIL import Image
from PIL.Image import Image as ImageType

from marqo_amd.s2_inference.errors import ChunkerError, ChunkerMethodProcessError
from marqo_amd.s2_inference.image_input import format_and_load_CLIP_image, pil_to_pixels, pil_to_rgb_u8

_local = threading.local()


def get_default_size() -> Tuple[int, int]:
    return (240, 240)


def str2bool(string: str) -> bool:
    return string.lower() in ("true", "1", "t")   # image_utils.py:204-213 (pinned by tests/test_ref_parity.py)


def rescale_box(box, from_size: Tuple, to_size: Tuple) -> List[float]:
    fy, fx = to_size[1] / from_size[1], to_size[0] / from_size[0]
    x1, y1, x2, y2 = box
    return [x1 * fx, y1 * fy, x2 * fx, y2 * fy]


def generate_boxes(image_size: Tuple[int, int], hn: int, wn: int, overlap: bool = False) -> List[Tuple]:
    """grid of (x1, y1, x2, y2) integer boxes; cells that would exceed the image are skipped; `overlap` adds a box shifted
    by half a cell after every grid cell (image_utils.py:165-202)."""
    img_width, img_height = image_size
    height, width = img_height // hn, img_width // wn
    bboxes = []
    for i in range(0, img_height, height):
        for j in range(0, img_width, width):
            p1, p2 = j + width, i + height
            if p1 > img_width or p2 > img_height:
                continue
            bboxes.append((j, i, p1, p2))
            if overlap:
                p3, p4 = p1 + width // 2, p2 + height // 2
                if p3 > img_width or p4 > img_height:
                    continue
                bboxes.append((j + width // 2, i + height // 2, p3, p4))
    return bboxes


def patchify_image(image: ImageType, bboxes) -> List[ImageType]:
    return [image.crop(bb) for bb in bboxes]


def _process_patch_method(method: str):
    """'simple', 'simple?hn=3', 'overlap?hn=3&wn=4' -> (method, params)"""
    req = urlparse(method)
    params = dict()
    if len(req.query) == 0:
        return req.path, params
    try:
        params = dict(x.split("=") for x in req.query.split("&"))
    except Exception:
        raise ChunkerMethodProcessError(message=f"could not pass parameters for string {req.query} from full path {method}")
    return req.path, params


def _preprocessor(device: str, size: int = 224):
    from marqo_amd.engine.preprocess import ImagePreprocessor
    key = (device, size)
    cache = getattr(_local, "pre", None)
    if cache is None:
        cache = _local.pre = {}
    if key not in cache:
        cache[key] = ImagePreprocessor(device, size)
    return cache[key]


class PatchifySimple:
    def __init__(self, size: Tuple = (512, 512), hn: int = 3, wn: int = 3, overlap: bool = False, device: str = "cuda", **kwargs):
        self.size, self.hn, self.wn, self.overlap, self.device = size, hn, wn, overlap, device

    def infer(self, image: Union[str, ImageType]):
        self.image = format_and_load_CLIP_image(image, {})
        self.original_size = self.image.size
        u8 = _preprocessor(self.device).resize_u8([pil_to_pixels(self.image)], self.size[1], self.size[0])
        self.image_resized = Image.fromarray(u8[0].cpu().numpy(), "RGB")
        self.bboxes_simple = generate_boxes(self.size, self.hn, self.wn, overlap=self.overlap)

    def process(self):
        self.bboxes = [(0, 0, self.size[0], self.size[1])] + self.bboxes_simple
        self.patches = patchify_image(self.image_resized, self.bboxes)
        self.bboxes_orig = [rescale_box(bb, self.size, self.original_size) for bb in self.bboxes]


def chunk_image(image: Union[str, ImageType], device: str, method: str, size=get_default_size()):
    HN = WN = 3
    if method in [None, "none", "", "None", " "]:
        if isinstance(image, str):
            return [image], [image]
        elif isinstance(image, ImageType):
            return [image], [(0, 0, image.size[0], image.size[1])]
        raise TypeError(f"only pointers to an image or a PIL image are allowed. received {type(image)}")
    method, params = _process_patch_method(method)
    hn, wn = int(params.get("hn", HN)), int(params.get("wn", WN))
    if method == "simple":
        patch = PatchifySimple(size=size, hn=hn, wn=wn, device=device)
    elif method == "overlap":
        patch = PatchifySimple(size=size, hn=hn, wn=wn, overlap=True, device=device)
    elif method in ["fastercnn", "frcnn", "marqo-yolo", "yolox", "dino-v1", "dino-v2", "dino/v1", "dino/v2"]:
        raise ChunkerError(f"patch method {method!r} needs a detector model, which the marqo_amd engine does not provide")
    else:
        raise ValueError(f"unexpected image chunking type. found {method}")
    try:
        patch.infer(image)
        patch.process()
    except PIL.UnidentifiedImageError as e:
        raise ChunkerError from e
    return patch.patches, patch.bboxes_orig


def chunk_images_to_tensors(images: List[Union[str, ImageType, np.ndarray]], model, method: str = "simple"):
    """Fused K11 path for a loaded CLIP-family engine model: -> (Tensor [n, count, 3, S, S] fp32 on device, boxes [n, count, 4])."""
    method, params = _process_patch_method(method)
    if method not in ("simple", "overlap"):
        raise ValueError(f"unexpected image chunking type. found {method}")
    hn, wn = int(params.get("hn", 3)), int(params.get("wn", 3))
    raw = [pil_to_pixels(format_and_load_CLIP_image(i, {})) if not isinstance(i, np.ndarray) else i for i in images]
    pre = model._pre()
    u8, boxes = pre.chunk_grid_u8(raw, hn, wn, method == "overlap")
    t = pre.to_tensor_normalize(u8)
    return t.reshape(len(raw), -1, *t.shape[1:]), boxes
