"""`marqo.s2_inference.clip_utils` by name (src/marqo/s2_inference/clip_utils.py): the module the reference's add_documents path imports
for image loading (`tensor_search/add_docs.py:23,139-143,187-198`: `clip_utils.load_image_from_path`, `clip_utils._is_image`) and its
loader map for the OpenAI-CLIP loaders (`model_registry.py:2133-2145`: CLIP, FP16_CLIP, MULTILINGUAL_CLIP).  Everything lives in
image_input.py / open_clip_model.py / s2_inference.py; this module only gives the reference's import path a target, so that
`from marqo_amd.s2_inference import clip_utils` is a one-line swap like the `s2_inference` one."""
from marqo_amd.engine.archs import OPENAI_DATASET_MEAN, OPENAI_DATASET_STD  # noqa: F401
from marqo_amd.s2_inference.image_input import (DEFAULT_HEADERS, _is_image, download_image_from_url, format_and_load_CLIP_image,  # noqa: F401
                                                format_and_load_CLIP_images, get_allowed_image_types, load_image_from_path)
from marqo_amd.s2_inference.s2_inference import encode_url, validate_url  # noqa: F401

HF_HUB_PREFIX = "hf-hub:"
MARQO_OPEN_CLIP_REGISTRY_PREFIX = "open_clip/"


def _convert_image_to_rgb(image):
    """clip_utils.py:43-45"""
    return image.convert("RGB")


def __getattr__(name):
    # the loader classes import this package's engine (HIP library) — resolved on first use, so that the image helpers above stay importable
    # in a process that only downloads / decodes media
    if name in ("CLIP", "FP16_CLIP", "MULTILINGUAL_CLIP", "get_multilingual_clip_properties"):
        from marqo_amd.s2_inference import open_clip_model
        return getattr(open_clip_model, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
