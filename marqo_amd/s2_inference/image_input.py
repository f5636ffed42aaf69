"""Typing and loading of image inputs on the host (before pixels reach the GPU).

Mirrors src/marqo/core/inference/embedding_models/image_download.py:24-178: `_is_image` decides from the FIRST
element only; strings are local files or URLs; ndarray -> RGB image; PIL passes; Tensors pass untouched (they are
already-preprocessed [3, S, S] tensors from `.preprocess`).
"""
from __future__ import annotations

import os
from io import BytesIO
from typing import List, Union
from urllib.parse import urlparse

import numpy as np
import torch
from PIL import Image, UnidentifiedImageError
from PIL.Image import Image as ImageType

from marqo_amd.s2_inference.errors import ImageDownloadError, InternalError

DEFAULT_HEADERS = {"User-Agent": "Marqobot/1.0"}


def get_allowed_image_types():
    return {".jpg", ".png", ".bmp", ".jpeg"}


def _looks_like_url(s: str) -> bool:
    """stand-in for `validators.url` (not installed): scheme http/https/ftp(s) + a network location, no spaces."""
    if any(c.isspace() for c in s):
        return False
    try:
        u = urlparse(s)
    except ValueError:
        return False
    return u.scheme in ("http", "https", "ftp", "ftps") and bool(u.netloc) and "." in u.netloc or \
        (u.scheme in ("http", "https") and u.netloc.split(":")[0] == "localhost")


def _is_image(inputs) -> bool:
    _allowed = get_allowed_image_types()
    if isinstance(inputs, list):
        if len(inputs) == 0:
            raise UnidentifiedImageError("received empty list, expected at least one element.")
        thing = inputs[0]
    else:
        thing = inputs
    if isinstance(thing, str):
        _, extension = os.path.splitext(thing.lower())
        if extension in _allowed:
            return True
        if os.path.isfile(thing):
            raise UnidentifiedImageError(
                f"local file [{thing}] extension {extension} does not match allowed file types of {_allowed}")
        return _looks_like_url(thing)
    elif isinstance(thing, (ImageType, np.ndarray, torch.Tensor)):
        return True
    raise UnidentifiedImageError(f"expected type Image or str for inputs but received type {type(thing)}")


def download_image_from_url(image_path: str, image_download_headers: dict, timeout_ms: int = 3000) -> BytesIO:
    """The reference uses pycurl (image_download.py:166-214); `requests` is what this image has."""
    if not isinstance(timeout_ms, int):
        raise InternalError(f"timeout must be an integer but received {timeout_ms} of type {type(timeout_ms)}")
    import requests
    try:
        url = requests.utils.requote_uri(image_path)
    except UnicodeEncodeError as e:
        raise ImageDownloadError(f"Marqo encountered an error when downloading the image url {image_path}. "
                                 f"The url could not be encoded properly. Original error: {e}")
    headers = DEFAULT_HEADERS.copy()
    headers.update(image_download_headers or {})
    try:
        resp = requests.get(url, headers=headers, timeout=timeout_ms / 1000.0, allow_redirects=True)
    except requests.RequestException as e:
        raise ImageDownloadError(f"Marqo encountered an error when downloading the image url {image_path}. "
                                 f"The original error is: {e}")
    if resp.status_code != 200:
        raise ImageDownloadError(f"image url `{image_path}` returned {resp.status_code}")
    return BytesIO(resp.content)


def load_image_from_path(image_path: str, image_download_headers: dict, timeout_ms=3000, metrics_obj=None) -> ImageType:
    """clip_utils.py:94-131 / image_download.py:130-163: a local file, else a URL; `metrics_obj` (the caller's RequestMetrics,
    add_docs.py:140-143) times the download under the key the reference uses"""
    if os.path.isfile(image_path):
        return Image.open(image_path)
    if _looks_like_url(image_path):
        if metrics_obj is not None:
            metrics_obj.start(f"image_download.{image_path}")
        try:
            return Image.open(download_image_from_url(image_path, image_download_headers, timeout_ms))
        except ImageDownloadError as e:
            raise UnidentifiedImageError(str(e)) from e
        finally:
            if metrics_obj is not None:
                metrics_obj.stop(f"image_download.{image_path}")
    raise UnidentifiedImageError(f"Input str of {image_path} is not a local file or a valid url.")


def format_and_load_CLIP_image(image, image_download_headers: dict) -> Union[ImageType, torch.Tensor]:
    if isinstance(image, str):
        return load_image_from_path(image, image_download_headers)
    if isinstance(image, np.ndarray):
        return Image.fromarray(image.astype("uint8"), "RGB") if image.ndim == 3 else Image.fromarray(image.astype("uint8")).convert("RGB")
    if isinstance(image, (torch.Tensor, ImageType)):
        return image
    raise UnidentifiedImageError(f"input of type {type(image)} did not match allowed types of str, np.ndarray, ImageType, Tensor")


def format_and_load_CLIP_images(images: List, image_download_headers: dict) -> List:
    if not isinstance(images, list):
        raise TypeError(f"expected list but received {type(images)}")
    return [format_and_load_CLIP_image(i, image_download_headers) for i in images]


def pil_to_pixels(img: ImageType):
    """PIL image -> what the GPU preprocessing packs, by image mode (engine/preprocess.py::pil_pixels): RGB images as a zero-copy RGBX
    view of Pillow's own memory where this Pillow exports one (repacked to RGB on the device) or uint8 [H, W, 3]; translucent RGBA / LA
    images and palette / bilevel images in containers that make the device resize them the way Pillow resizes those modes (the
    reference converts to RGB AFTER Resize / CenterCrop, clip_utils.py:61-64)."""
    from marqo_amd.engine.preprocess import pil_pixels
    return pil_pixels(img)


def pil_to_rgb_u8(img: ImageType) -> np.ndarray:
    """PIL image -> uint8 [H, W, 3], flattened to RGB up front (chunk grids, callers that want plain arrays).  The reference converts to
    RGB AFTER Resize / CenterCrop (clip_utils.py:61-64); for RGB and L inputs the order is immaterial, for translucent or palette
    images use pil_to_pixels, which keeps the mode-dependent resize."""
    if img.mode != "RGB":
        img = img.convert("RGB")
    return np.asarray(img)
