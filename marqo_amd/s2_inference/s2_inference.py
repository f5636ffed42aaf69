"""`vectorise()` and the model cache: the API surface tensor_search's add_documents / search / embed paths call.

Mirrors src/marqo/s2_inference/s2_inference.py function by function (same names, argument meaning and error
behaviour; citations inline) so the reference's callers and tests drop onto it.  What changes is underneath:

  * loaders in MODEL_PROPERTIES['loaders'] are the marqo_amd engine classes (HIP towers), not torch modules;
  * `_encode_without_cache` hands engine models the WHOLE request in one `encode` call — they micro-batch on the
    device by token rows — instead of the fixed 16-item Python loop (s2_inference.py:135-146).  Models that do
    not declare `supports_dynamic_batching` (the `random` fake, third-party loaders) keep the reference loop and
    `MARQO_MAX_VECTORISE_BATCH_SIZE`;
  * `vectorise_ndarray` is an additional zero-copy exit for callers that can take `np.ndarray` instead of
    `List[List[float]]`.
"""
from __future__ import annotations

import datetime
import logging
import threading
from typing import Any, Dict, List, Literal, Optional, Tuple, Union  # noqa: F401  (also part of the module's namespace, see below)

import numpy as np
import torch
from numpy import ndarray  # noqa: F401
from PIL import UnidentifiedImageError
from PIL.Image import Image
from PIL.Image import Image as ImageType  # noqa: F401
from torch import FloatTensor, Tensor  # noqa: F401

from marqo_amd.s2_inference import configs
from marqo_amd.s2_inference.configs import (get_default_normalization, get_default_seq_length, read_env_vars_and_defaults,
                                            read_env_vars_and_defaults_ints)
from marqo_amd.s2_inference.enums import AvailableModelsKey, EnvVars, Modality, ModelType
from marqo_amd.s2_inference.errors import (ConfigurationError, InternalError, InvalidModelPropertiesError, MediaDownloadError,  # noqa: F401
                                           ModelCacheManagementError, ModelDownloadError, ModelLoadError,
                                           ModelNotInCacheError, UnknownModelError, VectoriseError)
from marqo_amd.s2_inference import coalesce as _coalesce
from marqo_amd.s2_inference.inference_cache import MarqoInferenceCache
from marqo_amd.s2_inference.model_registry import load_model_properties

constants = configs   # (the reference keeps PATCH_MODELS / PREPROCESS_IMAGE_MODEL_LIST in s2_inference/constants.py)


def get_logger(name):
    """s2_inference/logger.py:3-17 — other reference modules import it FROM this module (processing/image.py:9, image_utils.py:8,
    reranking/model_utils.py:19: the reference's s2_inference.py star-imports its helpers, so they are part of the module's namespace a
    drop-in has to keep: get_logger, the typing / tensor type names of s2_inference/types.py, _float_tensor_to_list, _nd_array_to_list)"""
    lg = logging.getLogger(name)
    lg.setLevel(logging.INFO)
    return lg


logger = logging.getLogger(__name__)

# {"model_cache_key": {"model": obj, "most_recently_used_time": t, "model_size": GB}}
_available_models: Dict[str, Dict[str, Any]] = dict()
lock = threading.Lock()
MODEL_PROPERTIES = load_model_properties()
_marqo_inference_cache = MarqoInferenceCache(
    cache_size=read_env_vars_and_defaults_ints(EnvVars.MARQO_INFERENCE_CACHE_SIZE),
    cache_type=read_env_vars_and_defaults(EnvVars.MARQO_INFERENCE_CACHE_TYPE))


class DefaultEncoder:
    """drops `modality` before calling the model (multimodal_model_load.py:124-129)"""

    def __init__(self, model):
        self.model = model

    def encode(self, content, modality, **kwargs):
        return self.model.encode(content, **kwargs)


def get_encoder(model):
    return DefaultEncoder(model)


def generate_batches(seq, batch_size: int):
    """tensor_search/utils.py:334-340"""
    if batch_size < 1:
        raise ValueError("Batch size must be greater than 0")
    for i in range(0, len(seq), batch_size):
        yield seq[i:i + batch_size]


def validate_url(url) -> bool:
    """clip_utils.py:134-144 (`validators.url` there; the same scheme / host test as image_input._is_image here)"""
    from marqo_amd.s2_inference.image_input import _looks_like_url
    return isinstance(url, str) and (_looks_like_url(url) or _looks_like_url(encode_url(url)))


def encode_url(url: str) -> str:
    """clip_utils.py:196-214: percent-encoding as requests.utils.requote_uri"""
    try:
        import requests
        return requests.utils.requote_uri(url)
    except ImportError:
        from urllib.parse import quote
        return quote(url, safe="!#$%&'()*+,/:;=?@[]~")


_IMAGE_EXT, _VIDEO_EXT, _AUDIO_EXT = ("jpg", "jpeg", "png", "gif", "webp"), ("mp4", "avi", "mov"), ("mp3", "wav", "ogg")


def _sniff_mime_family(head: bytes) -> Optional[str]:
    """'image' / 'video' / 'audio' / None from the first bytes of a file — the container signatures behind the MIME families the reference
    asks python-magic for (multimodal_model_load.py:171-177,190-198; libmagic is not a dependency of this engine)"""
    if head.startswith((b"\x89PNG\r\n\x1a\n", b"\xff\xd8\xff", b"GIF87a", b"GIF89a", b"BM")) or head[:4] in (b"II*\x00", b"MM\x00*"):
        return "image"
    if head[:4] == b"RIFF" and len(head) >= 12:
        return {b"WEBP": "image", b"AVI ": "video", b"WAVE": "audio"}.get(head[8:12])
    if head[4:8] == b"ftyp":
        brand = head[8:12]
        return "image" if brand in (b"avif", b"heic", b"heix", b"mif1") else "audio" if brand in (b"M4A ", b"M4B ") else "video"
    if head.startswith(b"\x1aE\xdf\xa3"):
        return "video"                                     # Matroska / WebM
    if head.startswith((b"ID3", b"OggS", b"fLaC")) or (len(head) > 1 and head[0] == 0xFF and head[1] & 0xE0 == 0xE0):
        return "audio"
    return None


def infer_modality(content, media_download_headers: Optional[dict] = None, timeout_ms: int = 3000) -> Modality:
    """multimodal_model_load.py:148-203, the function the reference's search and add_documents paths import FROM this module
    (tensor_search.py:74, add_docs.py:24-25).  A string that is not a URL is text; a URL is classified by its extension (image / video /
    audio lists of the reference), else by the first 10 KB of what it serves; bytes by their signature; anything else is text.  Video and
    audio are only NAMED here (the callers answer them with UnsupportedModalityError for every model family this engine runs)."""
    if isinstance(content, str):
        if not validate_url(content):
            return Modality.TEXT
        encoded = encode_url(content)
        ext = encoded.split(".")[-1].lower()
        if ext in _IMAGE_EXT:
            return Modality.IMAGE
        if ext in _VIDEO_EXT:
            return Modality.VIDEO
        if ext in _AUDIO_EXT:
            return Modality.AUDIO
        try:
            import requests
        except ImportError:
            return Modality.TEXT
        try:
            # (connect, read) timeouts and the request's media download headers: an unresponsive host must not park a search / add_documents
            # request thread for ever (the reference's probe has neither, multimodal_model_load.py:171-177)
            resp = requests.get(encoded, stream=True, timeout=(timeout_ms / 1000.0, timeout_ms / 1000.0), headers=media_download_headers or None)
            try:
                head = b""
                for chunk in resp.iter_content(chunk_size=8192):
                    head += chunk
                    if len(head) >= 10240:
                        break
            finally:
                resp.close()
        except requests.exceptions.RequestException as e:
            raise MediaDownloadError(f"Error downloading media file {content}: {e}") from e
        except IOError as e:
            raise MediaDownloadError(f"IO error while processing {encoded}: {e}") from e
        family = _sniff_mime_family(head)
        return {"image": Modality.IMAGE, "video": Modality.VIDEO, "audio": Modality.AUDIO}.get(family, Modality.TEXT)
    if isinstance(content, bytes):
        return {"image": Modality.IMAGE, "video": Modality.VIDEO, "audio": Modality.AUDIO}.get(_sniff_mime_family(content), Modality.TEXT)
    return Modality.TEXT


# =============================================================================================================
def vectorise(model_name: str, content: Union[str, List[str], List[Image], List[bytes]], model_properties: dict = None,
              device: str = None, normalize_embeddings: bool = get_default_normalization(), model_auth=None,
              enable_cache: bool = False, modality: Modality = Modality.TEXT, **kwargs) -> List[List[float]]:
    """s2_inference.py:48-69"""
    if not device:
        raise InternalError(message="vectorise (internal function) cannot be called without setting device!")
    validated_model_properties = validate_model_properties(model_name, model_properties)
    model_cache_key = _create_model_cache_key(model_name, device, validated_model_properties)
    _update_available_models(model_cache_key, model_name, validated_model_properties, device, normalize_embeddings,
                             model_auth=model_auth)
    model = _available_models[model_cache_key][AvailableModelsKey.model]
    if _marqo_inference_cache.is_enabled() and enable_cache:
        return _vectorise_with_cache(model, model_cache_key, content, normalize_embeddings, modality, **kwargs)
    return _vectorise_without_cache(model_cache_key, content, normalize_embeddings, modality, **kwargs)


def vectorise_ndarray(model_name: str, content, model_properties: dict = None, device: str = None,
                      normalize_embeddings: bool = get_default_normalization(), model_auth=None,
                      modality: Modality = Modality.TEXT, **kwargs) -> np.ndarray:
    """Engine extension: same as `vectorise` (cache off) but returns the fp32 [N, D] ndarray, skipping the
    N*D Python-float materialisation of `_convert_vectorized_output`."""
    if not device:
        raise InternalError(message="vectorise (internal function) cannot be called without setting device!")
    props = validate_model_properties(model_name, model_properties)
    key = _create_model_cache_key(model_name, device, props)
    _update_available_models(key, model_name, props, device, normalize_embeddings, model_auth=model_auth)
    out = _encode_to_array(key, content, normalize_embeddings, modality, **kwargs)
    return out[np.newaxis, :] if out.ndim == 1 else out


def vectorise_device(model_name: str, content, model_properties: dict = None, device: str = None,
                     normalize_embeddings: bool = get_default_normalization(), model_auth=None,
                     modality: Modality = Modality.TEXT, **kwargs) -> torch.Tensor:
    """Engine extension: same flow as `vectorise` (cache off) but the fp32 [N, D] result STAYS IN HBM as a
    torch.Tensor on `device` — no D2H, no Python floats.  The bulk-ingest path (marqo_amd.ingest.BulkVectoriser) hands these
    shards straight to the RCCL all_gather.  Models that cannot produce device tensors (random / no_model fakes) are copied up."""
    if not device:
        raise InternalError(message="vectorise (internal function) cannot be called without setting device!")
    props = validate_model_properties(model_name, model_properties)
    key = _create_model_cache_key(model_name, device, props)
    _update_available_models(key, model_name, props, device, normalize_embeddings, model_auth=model_auth)
    model = _available_models[key][AvailableModelsKey.model]
    if getattr(model, "supports_dynamic_batching", False) is True:
        kwargs["return_device"] = True
    out = _encode_to_array(key, content, normalize_embeddings, modality, **kwargs)
    if not isinstance(out, torch.Tensor):
        out = torch.from_numpy(np.ascontiguousarray(out, dtype=np.float32))
        if str(device).startswith("cuda") and torch.cuda.is_available():
            out = out.to(device)
    return out[None, :] if out.ndim == 1 else out


def _vectorise_with_cache(model, model_cache_key, content, normalize_embeddings, modality, **kwargs):
    """s2_inference.py:72-84"""
    if isinstance(content, str):
        vectorised = _marqo_inference_cache.get(model_cache_key, content)
        if vectorised is None:
            vectorised = _encode_without_cache(model_cache_key, content, normalize_embeddings, modality, **kwargs)
            _marqo_inference_cache.set(model_cache_key, content, vectorised[0])
        else:
            vectorised = _convert_cached_embeddings_to_output(vectorised)
        return vectorised
    elif isinstance(content, list):
        return _vectorise_list_with_cache(model, model_cache_key, content, normalize_embeddings, modality, **kwargs)
    raise TypeError(f"Unsupported content type: {type(content).__name__}")


def _vectorise_list_with_cache(model, model_cache_key, content, normalize_embeddings, modality, **kwargs):
    """s2_inference.py:87-115: only str items are cached; misses are encoded together and hits re-inserted at their
    original positions in ascending index order."""
    contents_to_vectorise, cached_output = [], []
    for loc, item in enumerate(content):
        if isinstance(item, str):
            vectorised = _marqo_inference_cache.get(model_cache_key, item)
            if vectorised is None:
                contents_to_vectorise.append(item)
            else:
                cached_output.append((loc, vectorised))
        else:
            contents_to_vectorise.append(item)
    if contents_to_vectorise:
        vectorised_outputs = _encode_without_cache(model_cache_key, contents_to_vectorise, normalize_embeddings, modality, **kwargs)
        for item, out in zip(contents_to_vectorise, vectorised_outputs):
            if isinstance(item, str):
                _marqo_inference_cache.set(model_cache_key, item, out)
        for loc, cached_vector in cached_output:
            vectorised_outputs.insert(loc, cached_vector)
    else:
        vectorised_outputs = [vector for _, vector in cached_output]
    return vectorised_outputs


def _vectorise_without_cache(model_cache_key, content, normalize_embeddings, modality, **kwargs) -> List[List[float]]:
    return _encode_without_cache(model_cache_key, content, normalize_embeddings, modality, **kwargs)


def _coalesce_key(model_cache_key, modality, normalize, infer, kwargs):
    """calls may share an engine call only when everything but the content is equal; unhashable keyword arguments (download headers as a
    dict are made hashable, anything stranger opts the call out)"""
    try:
        kw = tuple(sorted((k, tuple(sorted(v.items())) if isinstance(v, dict) else v) for k, v in kwargs.items()))
        hash(kw)
    except TypeError:
        return None
    return (model_cache_key, str(modality), bool(normalize), bool(infer), kw)


def _encode_to_array(model_cache_key: str, content, normalize_embeddings: bool, modality, **kwargs):
    """s2_inference.py:123-156 up to (not including) the list conversion."""
    try:
        model = _available_models[model_cache_key][AvailableModelsKey.model]
        encoder = get_encoder(model)
        if isinstance(content, str):
            vectorised = model.encode(content, normalize=normalize_embeddings, modality=modality, **kwargs)
        elif isinstance(content, torch.Tensor):
            vectorised = model.encode(content, normalize=normalize_embeddings, modality=modality, **kwargs)
        else:
            vector_batches = []
            on_device = bool(kwargs.get("return_device"))
            batch_size = _get_max_vectorise_batch_size()  # validated even when the engine batches dynamically
            dynamic = getattr(model, "supports_dynamic_batching", False) is True and len(content) > 0   # (`is True`: a mock model answers truthy to any attribute)
            if dynamic:
                # one call: the engine micro-batches by token rows on the device.  Non-text items (decoded images: ~MBs of pinned
                # staging + HBM each) are still bounded per call, by MARQO_AMD_MAX_ITEMS_PER_ENCODE (default 1024), so a request of
                # thousands of large images cannot stage tens of GB at once.
                batch_size = len(content)
                if not isinstance(content[0], str):
                    batch_size = min(batch_size, max(1, read_env_vars_and_defaults_ints(EnvVars.MARQO_AMD_MAX_ITEMS_PER_ENCODE)))
            # small concurrent calls for the same (model, modality, arguments) share ONE engine call (coalesce.py; default: calls of <= 16
            # items, MARQO_AMD_COALESCE_US=0 turns it off) — what feeds the GPU when an unmodified Marqo vectorises per document and field from
            # 8 + 8 request threads.  A lone caller is never delayed.
            coalesce_window = _coalesce.window_for(len(content)) if dynamic else 0.0
            for batch in generate_batches(content, batch_size=batch_size):
                if modality is None:
                    modality = infer_modality(batch[0] if isinstance(batch[0], (str, bytes)) else batch)
                infer = kwargs.pop("infer", False if modality == Modality.TEXT else True)
                ckey = _coalesce_key(model_cache_key, modality, normalize_embeddings, infer, kwargs) if coalesce_window > 0 else None
                if ckey is not None and not _coalesce.explicit() and _coalesce.fetches_content(batch, modality == Modality.TEXT):
                    ckey = None         # image URLs: every request thread keeps downloading its own (coalesce.fetches_content)
                if ckey is not None and modality == Modality.TEXT and not _coalesce.explicit():
                    # a text tower with a native request queue (engine/native_queue.py, csrc/queue.hip) merges concurrent small calls itself, on
                    # worker threads outside the interpreter: this thread tokenises its own texts and blocks in ONE foreign call
                    takes = getattr(model, "native_queue_takes", None)
                    if callable(takes) and takes(batch) is True:
                        ckey = None
                elif ckey is not None and modality == Modality.IMAGE and not _coalesce.explicit() and not kwargs.get("return_device"):
                    takes = getattr(model, "native_queue_takes_images", None)     # (... and so does an image tower's, for the tensors `.preprocess` returns)
                    if callable(takes) and takes(batch) is True:
                        ckey = None
                if ckey is not None:
                    def run_merged(items, _m=modality, _i=infer, _kw=dict(kwargs)):
                        return encoder.encode(items, modality=_m, normalize=normalize_embeddings, infer=_i, **_kw)
                    encoded_batch = _coalesce.get_coalescer().submit(ckey, list(batch), run_merged, coalesce_window, _coalesce.max_items())
                else:
                    encoded_batch = encoder.encode(batch, modality=modality, normalize=normalize_embeddings, infer=infer, **kwargs)
                vector_batches.append(encoded_batch if on_device and isinstance(encoded_batch, torch.Tensor)
                                      else _convert_tensor_to_numpy(encoded_batch))
            if not vector_batches or all(len(b) == 0 for b in vector_batches):
                raise RuntimeError(f"Vectorise created an empty list of batches! Content: {content}")
            if len(vector_batches) == 1:
                vectorised = vector_batches[0]
            elif on_device and all(isinstance(b, torch.Tensor) for b in vector_batches):
                vectorised = torch.cat(vector_batches, dim=0)
            else:
                vectorised = np.concatenate([_convert_tensor_to_numpy(b) for b in vector_batches], axis=0)
    except (UnidentifiedImageError, OSError) as e:
        if isinstance(e, UnidentifiedImageError) or "image file is truncated" in str(e):
            raise VectoriseError(f"Could not process given image: {content}. Original Error message: {e}") from e
        raise e
    return vectorised


def _encode_without_cache(model_cache_key: str, content, normalize_embeddings: bool, modality, **kwargs) -> List[List[float]]:
    return _convert_vectorized_output(_encode_to_array(model_cache_key, content, normalize_embeddings, modality, **kwargs))


def get_available_models() -> Dict:
    return _available_models


def get_marqo_inference_cache() -> MarqoInferenceCache:
    return _marqo_inference_cache


def is_preprocess_image_model(model_properties: dict = None) -> bool:
    """s2_inference.py:180-185"""
    return model_properties.get("type", None) in configs.PREPROCESS_IMAGE_MODEL_LIST


def load_multimodal_model_and_get_preprocessors(model_name: str, model_properties: Optional[dict] = None, device: Optional[str] = None,
                                                model_auth=None, normalize_embeddings: bool = get_default_normalization()
                                                ) -> Tuple[Any, Dict[str, Optional[Any]]]:
    """s2_inference.py:193-235: loads (or renews) the model and returns its per-modality preprocessors; `image` is the model's
    `.preprocess` (PIL -> Tensor[3, S, S]) for clip / open_clip types, else None."""
    if not device:
        raise InternalError(message="vectorise (internal function) cannot be called without setting device!")
    model_cache_key = _create_model_cache_key(model_name, device, model_properties)
    _update_available_models(model_cache_key, model_name, model_properties, device, normalize_embeddings, model_auth=model_auth)
    model = _available_models[model_cache_key][AvailableModelsKey.model]
    preprocessors = {
        "image": getattr(model, "preprocess", None) if is_preprocess_image_model(model_properties) else None,
        "video": None, "audio": None, "text": None,
    }
    return model, preprocessors


def _get_max_vectorise_batch_size() -> int:
    """s2_inference.py:239-257"""
    max_batch_size_value = read_env_vars_and_defaults(EnvVars.MARQO_MAX_VECTORISE_BATCH_SIZE)
    validation_error_msg = ("Could not properly read env var `MARQO_MAX_VECTORISE_BATCH_SIZE`. "
                            "`MARQO_MAX_VECTORISE_BATCH_SIZE` must be an int greater than or equal to 1.")
    try:
        batch_size = int(max_batch_size_value)
    except (ValueError, TypeError) as e:
        msg = f"`{validation_error_msg} Current value: `{max_batch_size_value}`. Reason: {e}"
        logger.error(msg)
        raise ConfigurationError(msg) from e
    if batch_size < 1:
        msg = f"`{validation_error_msg} Current value: `{max_batch_size_value}`."
        logger.error(msg)
        raise ConfigurationError(msg)
    return batch_size


def _create_model_cache_key(model_name: str, device: str, model_properties: dict = None) -> str:
    """s2_inference.py:260-283 (eject_model depends on this format)"""
    if model_properties is None:
        model_properties = dict()
    return (model_name + "||" + model_properties.get("name", "") + "||" + str(model_properties.get("dimensions", "")) + "||"
            + model_properties.get("type", "") + "||" + str(model_properties.get("tokens", "")) + "||" + device)


def _update_available_models(model_cache_key: str, model_name: str, validated_model_properties: dict, device: str,
                             normalize_embeddings: bool, model_auth=None) -> None:
    """s2_inference.py:286-337: single-flight load with REJECTION of concurrent loaders (not waiting)."""
    if model_cache_key not in _available_models:
        model_size = get_model_size(model_name, validated_model_properties)
        if lock.locked():
            raise ModelCacheManagementError("Request rejected, as this request attempted to update the model cache, while "
                                            "another request was updating the model cache at the same time. "
                                            "Please wait for 10 seconds and send the request again ")
        with lock:
            _validate_model_into_device(model_name, validated_model_properties, device, calling_func=_update_available_models.__name__)
            try:
                most_recently_used_time = datetime.datetime.now()
                _available_models[model_cache_key] = {
                    AvailableModelsKey.model: _load_model(model_name, validated_model_properties, device=device,
                                                          calling_func=_update_available_models.__name__, model_auth=model_auth),
                    AvailableModelsKey.most_recently_used_time: most_recently_used_time,
                    AvailableModelsKey.model_size: model_size,
                }
                logger.info(f"loaded {model_name} on device {device} with normalization={normalize_embeddings} at time={most_recently_used_time}.")
            except Exception as e:
                logger.error(f"Error loading model {model_name} on device {device} with normalization={normalize_embeddings}. \n"
                             f"Error message is {str(e)}")
                if isinstance(e, ModelDownloadError):
                    raise e
                raise ModelLoadError(
                    f"Unable to load model={model_name} on device={device} with normalization={normalize_embeddings}. "
                    f"If you are trying to load a custom model, please check that model_properties={validated_model_properties} "
                    f"is correct and Marqo has access to the weights file. Original error: {e}") from e
    else:
        most_recently_used_time = datetime.datetime.now()
        try:
            _available_models[model_cache_key][AvailableModelsKey.most_recently_used_time] = most_recently_used_time
        except KeyError as e:
            raise ModelNotInCacheError(f"Marqo cannot renew model {model_name} on device {device} with normalization={normalize_embeddings}. "
                                       f"Maybe another thread is updating the model cache at the same time."
                                       f"Please wait for 10 seconds and send the request again.\n") from e


def validate_model_properties(model_name: str, model_properties: dict) -> dict:
    """s2_inference.py:340-407"""
    if model_properties is not None:
        required_keys = []
        postfix = ("Marqo is loading the model with default type 'sbert' as the type was not provided."
                   if "type" not in model_properties else "")
        model_type = model_properties.get("type", None)
        if model_type in (None, ModelType.SBERT):
            required_keys = ["dimensions", "name"]
            for key, value in [("type", ModelType.SBERT.value), ("tokens", get_default_seq_length())]:
                if key not in model_properties:
                    model_properties[key] = value
        elif model_type in (ModelType.OpenCLIP, ModelType.CLIP):
            required_keys = ["name", "dimensions"]
        elif model_type in (ModelType.HF_MODEL, ModelType.HF_STELLA):
            required_keys = ["dimensions"]
        elif model_type in (ModelType.NO_MODEL,):
            required_keys = ["dimensions"]
            if not model_name == "no_model":
                raise InvalidModelPropertiesError(f"To use the 'no_model' feature, you must provide 'model = no_model' and "
                                                  f"'type = no_model', but received 'model = {model_name}' and 'type = {model_type}'.")
        elif model_type in (ModelType.Test, ModelType.Random, ModelType.MultilingualClip, ModelType.FP16_CLIP,
                            ModelType.SBERT_ONNX, ModelType.CLIP_ONNX):
            pass
        else:
            raise InvalidModelPropertiesError(
                "Invalid model type. Please check the model type in model_properties. Supported model types are "
                + ", ".join(f"'{t.value}'" for t in (ModelType.SBERT, ModelType.OpenCLIP, ModelType.CLIP, ModelType.HF_MODEL,
                                                     ModelType.HF_STELLA, ModelType.NO_MODEL, ModelType.Test, ModelType.Random,
                                                     ModelType.MultilingualClip, ModelType.FP16_CLIP, ModelType.SBERT_ONNX,
                                                     ModelType.CLIP_ONNX)))
        for key in required_keys:
            if key not in model_properties:
                raise InvalidModelPropertiesError(f"model_properties has missing key '{key}'. please update your model properties with "
                                                  f"required key `{key}`. {postfix}")
    else:
        model_properties = get_model_properties_from_registry(model_name)
    _validate_model_properties_dimension(model_properties.get("dimensions", None))
    return model_properties


def _validate_model_properties_dimension(dimensions: Optional[int]) -> None:
    if dimensions is None or not isinstance(dimensions, int) or dimensions < 1:
        raise InvalidModelPropertiesError(
            f"Invalid model properties: 'dimensions' must be a positive integer, but received {dimensions}.")


def _validate_model_into_device(model_name: str, model_properties: dict, device: str, calling_func: str = None) -> bool:
    """s2_inference.py:421-460: LRU ejection per device until the new model fits under the GB threshold."""
    if calling_func not in ["unit_test", "_update_available_models"]:
        raise RuntimeError("This function should only be called by `update_available_models` or `unit_test` for thread safeness.")
    model_size = get_model_size(model_name, model_properties)
    if _check_memory_threshold_for_model(device, model_size, calling_func=_validate_model_into_device.__name__):
        return True
    keys = [k for k in list(_available_models) if k.endswith(device)]
    for key in sorted(keys, key=lambda x: _available_models[x][AvailableModelsKey.most_recently_used_time]):
        logger.info(f"Eject model = `{key.split('||')[0]}` from device = `{device}` to save space for model = `{model_name}`.")
        del _available_models[key]
        if _check_memory_threshold_for_model(device, model_size, calling_func=_validate_model_into_device.__name__):
            return True
    if _check_memory_threshold_for_model(device, model_size, calling_func=_validate_model_into_device.__name__) is False:
        raise ModelCacheManagementError(
            f"Marqo CANNOT find enough space to load model = `{model_name}` in device = `{device}`.\n"
            f"Marqo tried to eject all the models on this device = `{device}` but still can't find enough space. \n"
            f"Please use a smaller model or increase the memory threshold.")


def _check_memory_threshold_for_model(device: str, model_size: Union[float, int], calling_func: str = None) -> bool:
    """s2_inference.py:463-500"""
    if calling_func not in ["unit_test", "_validate_model_into_device"]:
        raise RuntimeError(f"The function `{_check_memory_threshold_for_model.__name__}` should only be called by "
                           f"`unit_test` or `_validate_model_into_device` for threading safeness.")
    if device.startswith("cuda"):
        if torch.cuda.is_available():
            torch.cuda.synchronize(device)
            torch.cuda.empty_cache()
        used_memory = sum(_available_models[k].get("model_size", configs.DEFAULT_MODEL_SIZE) for k in _available_models if k.endswith(device))
        threshold = float(read_env_vars_and_defaults(EnvVars.MARQO_MAX_CUDA_MODEL_MEMORY))
    elif device.startswith("cpu"):
        used_memory = sum(_available_models[k].get("model_size", configs.DEFAULT_MODEL_SIZE) for k in _available_models if k.endswith("cpu"))
        threshold = float(read_env_vars_and_defaults(EnvVars.MARQO_MAX_CPU_MODEL_MEMORY))
    else:
        raise ModelCacheManagementError(f"Unable to check the device cache for device=`{device}`. The model loading will proceed"
                                        f"without device cache check. This might break down Marqo if too many models are loaded.")
    if model_size > threshold:
        raise ModelCacheManagementError(
            f"You are trying to load a model with size = `{model_size}` into device = `{device}`, which is larger than the device "
            f"threshold = `{threshold}`. Marqo CANNOT find enough space for the model. Please change the threshold by adjusting the "
            f"environment variables `MARQO_MAX_CUDA_MODEL_MEMORY` or `MARQO_MAX_CPU_MODEL_MEMORY`.")
    return (used_memory + model_size) < threshold


def get_model_size(model_name: str, model_properties: dict) -> Union[int, float]:
    """s2_inference.py:503-517: explicit model_size -> name substring -> type -> default"""
    if "model_size" in model_properties:
        return model_properties["model_size"]
    name_info = (model_name + model_properties.get("name", "")).lower().replace("/", "-")
    for name, size in configs.MODEL_NAME_SIZE_MAPPING.items():
        if name in name_info:
            return size
    return configs.MODEL_TYPE_SIZE_MAPPING.get(model_properties.get("type", None), configs.DEFAULT_MODEL_SIZE)


def _load_model(model_name: str, model_properties: dict, device: str, calling_func: str = None, model_auth=None) -> Any:
    """s2_inference.py:520-568"""
    if calling_func not in ["unit_test", "_update_available_models"]:
        raise RuntimeError(f"The function `{_load_model.__name__}` should only be called by "
                           f"`unit_test` or `_update_available_models` for threading safeness.")
    model_type = model_properties.get("type")
    loader = _get_model_loader(model_properties.get("name", None), model_properties)
    if model_type in (ModelType.OpenCLIP, ModelType.HF_MODEL, ModelType.HF_STELLA):
        model = loader(device=device, model_properties=model_properties, model_auth=model_auth)
    else:
        model = loader(model_properties.get("name", None), device=device, embedding_dim=model_properties["dimensions"],
                       model_properties=model_properties, model_auth=model_auth,
                       max_seq_length=model_properties.get("tokens", get_default_seq_length()))
    model.load()
    return model


def clear_loaded_models() -> None:
    _available_models.clear()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


def clear_marqo_inference_cache() -> None:
    if _marqo_inference_cache.is_enabled():
        _marqo_inference_cache.clear()


def get_model_properties_from_registry(model_name: str) -> dict:
    """s2_inference.py:599-620"""
    if model_name not in MODEL_PROPERTIES["models"]:
        raise UnknownModelError(f"Could not find model properties in model registry for model={model_name}. "
                                f"Model is not supported by default.")
    model_properties = MODEL_PROPERTIES["models"][model_name]
    validate_model_properties(model_name, model_properties)
    return model_properties


def _float_tensor_to_list(output) -> Union[List[List[float]], List[float]]:
    """s2_inference.py:650-661 (hard-coded to CPU)"""
    return output.detach().to("cpu").tolist()


def _nd_array_to_list(output) -> Union[List[List[float]], List[float]]:
    """s2_inference.py:664-674"""
    return output.tolist()


def _check_output_type(output) -> bool:
    """s2_inference.py:623-647 (soft check: only output[0][0] is inspected)"""
    if not isinstance(output, list):
        return False
    elif len(output) == 0:
        raise ValueError("received empty input")
    if not isinstance(output[0], list):
        return False
    elif len(output[0]) == 0:
        raise ValueError("received empty input")
    if not isinstance(output[0][0], (float, int)):
        return False
    return True


def _convert_tensor_to_numpy(output) -> np.ndarray:
    if isinstance(output, torch.Tensor):
        return output.to("cpu").detach().numpy()
    elif isinstance(output, np.ndarray):
        return output
    raise ValueError(f"Marqo received an unexpected output type=`{type(output).__name__}`from encode function.")


def _convert_cached_embeddings_to_output(cached_embeddings: List[float]) -> List[List[float]]:
    if not isinstance(cached_embeddings, list):
        raise TypeError(f"expected a list of floats but received {type(cached_embeddings)}")
    if not isinstance(cached_embeddings[0], float):
        raise TypeError(f"expected a list of floats but received {type(cached_embeddings[0])}")
    return [cached_embeddings, ]


def _convert_vectorized_output(output, fp16: bool = False) -> List[List[float]]:
    """s2_inference.py:705-749: anything -> List[List[float]]; 1-D gets a leading batch dim"""
    if _check_output_type(output):
        return output
    if isinstance(output, torch.Tensor):
        if output.ndim == 1:
            output = output.unsqueeze(0)
        output = output.detach().to("cpu").tolist()
    elif isinstance(output, np.ndarray):
        if output.ndim == 1:
            output = output[np.newaxis, :]
        output = output.tolist()
    elif isinstance(output, list):
        if isinstance(output[0], torch.Tensor):
            output = [o.detach().to("cpu").tolist() for o in output]
        elif isinstance(output[0], np.ndarray):
            output = [o.tolist() for o in output]
        else:
            raise TypeError(f"unsupported nested list with elements of type {type(output[0])}")
    else:
        raise TypeError(f"unsupported output type of {type(output)}")
    if fp16:
        output = np.array(output).astype(np.float16).tolist()
    if _check_output_type(output):
        return output
    raise TypeError(f"unable to convert input of type {type(output)} to a list of lists of floats")


def _get_model_loader(model_name: str, model_properties: dict) -> Any:
    model_type = model_properties["type"]
    if model_type not in MODEL_PROPERTIES["loaders"]:
        raise KeyError(f"model_name={model_name} for model_type={model_type} not in allowed model types")
    return MODEL_PROPERTIES["loaders"][model_type]


def eject_model(model_name: str, device: str):
    """s2_inference.py:774-798"""
    model_cache_key = None
    for key in _available_models.keys():
        if isinstance(key, str) and key.startswith(model_name) and key.endswith(device):
            model_cache_key = key
            break
    if model_cache_key is None:
        raise ModelNotInCacheError(f"The model_name `{model_name}` device `{device}` is not cached or found")
    if model_cache_key in _available_models:
        del _available_models[model_cache_key]
        if device.startswith("cuda") and torch.cuda.is_available():
            torch.cuda.empty_cache()
        return {"result": "success", "message": f"successfully eject model_name `{model_name}` from device `{device}`"}
    raise ModelNotInCacheError(f"The model_name `{model_name}` device `{device}` is not cached or found")
