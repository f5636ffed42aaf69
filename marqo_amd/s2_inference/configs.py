"""Defaults and environment plumbing (reference: s2_inference/configs.py:43-46, api/configs.py:27-38,
tensor_search/utils.py read_env_vars_and_defaults*)."""
import os
from typing import Optional

from marqo_amd.s2_inference.enums import EnvVars


_STATIC_DEFAULTS = None


def default_env_vars() -> dict:
    """the defaults; everything but the model directory is constant, and that one depends on two environment values only — vectorise() asks
    for a default on every call, so the dict is built once and only the directory is looked at again (it was a third of the per-request
    host time: expanduser + join + a dict literal per call)"""
    return dict(_defaults())


def _defaults() -> dict:
    global _STATIC_DEFAULTS
    sig = (os.environ.get("MARQO_ROOT_PATH"), os.environ.get("HOME"))
    if _STATIC_DEFAULTS is None or _STATIC_DEFAULTS[0] != sig:
        _STATIC_DEFAULTS = (sig, _build_defaults())
    return _STATIC_DEFAULTS[1]


def _build_defaults() -> dict:
    return {
        # reference defaults (api/configs.py:35-38): 4 GB per device, batch 16.  The 4 GB budget is far too
        # small for a 288 GB MI355X; the mechanism is kept and the default raised for cuda devices.
        EnvVars.MARQO_MAX_CPU_MODEL_MEMORY: 4,
        EnvVars.MARQO_MAX_CUDA_MODEL_MEMORY: 4,
        EnvVars.MARQO_MAX_VECTORISE_BATCH_SIZE: 16,
        EnvVars.MARQO_INFERENCE_CACHE_SIZE: 0,
        EnvVars.MARQO_INFERENCE_CACHE_TYPE: "LRU",
        EnvVars.MARQO_AMD_MODEL_DIR: os.path.join(os.environ.get("MARQO_ROOT_PATH", os.path.expanduser("~/.marqo")), "cache", "models"),
        EnvVars.MARQO_AMD_SYNTHETIC_WEIGHTS: "0",
        EnvVars.MARQO_AMD_MICRO_BATCH_ROWS: 65536,
        EnvVars.MARQO_AMD_MAX_ITEMS_PER_ENCODE: 1024,
    }


def read_env_vars_and_defaults(var: str) -> Optional[str]:
    """Env value if set and non-empty, else the default (tensor_search/utils.py:139-164)."""
    val = os.environ.get(var)
    if val is not None and val != "":
        return val
    return _defaults().get(var)


def read_env_vars_and_defaults_ints(var: str) -> Optional[int]:
    val = read_env_vars_and_defaults(var)
    if val is None or val == "":
        return None
    try:
        return int(val)
    except (ValueError, TypeError) as e:
        from marqo_amd.s2_inference.errors import ConfigurationError
        raise ConfigurationError(f"Unable to parse int from env var {var} with value {val}. Reason: {e}") from e


def get_default_normalization() -> bool:
    return True


def get_default_seq_length() -> int:
    return 128


# model-size accounting (constants.py:6-25), GB
MODEL_TYPE_SIZE_MAPPING = {"open_clip": 1, "clip": 1, "sbert": 0.7, "random": 0.1, "multilingual_clip": 5,
                           "clip_onnx": 1, "sbert_onnx": 0.7, "hf": 1}
MODEL_NAME_SIZE_MAPPING = {"vit-l-14": 1.5, "vit-g": 5, "vit-h": 5, "vit-bigg-14": 6}
DEFAULT_MODEL_SIZE = 0.66
PREPROCESS_IMAGE_MODEL_LIST = ["clip", "open_clip"]  # constants.py:31
PATCH_MODELS = {"simple", "overlap"}                  # the model-free chunkers of constants.py:27-29
