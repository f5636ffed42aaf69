"""Error hierarchy of the vectorise() path.

Same class names and inheritance as the reference (src/marqo/s2_inference/errors.py:4-72), plus the API-level exceptions the path raises
(src/marqo/api/exceptions.py:24-28,128-130,216-219,246-248: EnvVarError, ModelCacheManagementError, InternalError, ConfigurationError).

Names alone do not make the callers' `except` clauses work: the reference catches ITS OWN classes
(`except (s2_inference_errors.UnknownModelError, ..., s2_inference.ModelDownloadError)` in core/inference/tensor_fields_container.py:155-163
and tensor_search/tensor_search.py:1899-1911; the FastAPI handlers dispatch on `marqo.api.exceptions.MarqoWebError`), and its tests raise
its own classes INTO those clauses.  So when this engine is installed into the host application — the package `marqo` is importable — the
names below ARE the host's classes (identity: every isinstance / except works in both directions; message suffix, status code and error
code as the host defines them), and swapping the `s2_inference` import is the whole integration: tests/test_ref_parity.py runs the
reference's own callers and the reference's own unit tests on top of this package to check exactly that.  Without a host
(`MARQO_AMD_HOST_ERRORS=0`, or no `marqo` package: the GPU box, most tests) they are plain classes with the reference's constructor,
attributes and codes.
"""
import importlib
import os
from typing import Optional


def _host_module(name: str):
    if os.environ.get("MARQO_AMD_HOST_ERRORS", "1") == "0":
        return None
    try:
        return importlib.import_module(name)
    except Exception:  # noqa: BLE001 - no host application (or one that does not import here): stand-alone classes
        return None


_HOST_S2 = _host_module("marqo.s2_inference.errors")
_HOST_API = _host_module("marqo.api.exceptions")


def _host_class(module, name: str, base: Optional[type] = None):
    c = getattr(module, name, None) if module is not None else None
    return c if isinstance(c, type) and issubclass(c, base or Exception) else None


# ---- s2_inference errors -----------------------------------------------------------------------------------------------------------
class _StandAloneS2InferenceError(Exception):
    def __init__(self, message: Optional[str] = None) -> None:
        self.message = message
        super().__init__(self.message)


_StandAloneS2InferenceError.__name__ = _StandAloneS2InferenceError.__qualname__ = "S2InferenceError"
S2InferenceError = _host_class(_HOST_S2, "S2InferenceError") or _StandAloneS2InferenceError


def _s2(name: str) -> type:
    """the host's class of that name (when it derives from the host's S2InferenceError), else a stand-alone subclass"""
    host = _host_class(_HOST_S2, name, S2InferenceError)
    return host or type(name, (S2InferenceError,), {"__module__": __name__, "__doc__": f"src/marqo/s2_inference/errors.py: {name}"})


MediaMismatchError = _s2("MediaMismatchError")
ChunkerError = _s2("ChunkerError")
ChunkerMethodProcessError = _s2("ChunkerMethodProcessError")
VectoriseError = _s2("VectoriseError")
InvalidModelPropertiesError = _s2("InvalidModelPropertiesError")
UnknownModelError = _s2("UnknownModelError")
ModelLoadError = _s2("ModelLoadError")
ModelDownloadError = _s2("ModelDownloadError")
ModelNotInCacheError = _s2("ModelNotInCacheError")
IncompatibleModelDeviceError = _s2("IncompatibleModelDeviceError")
BatchInferenceSizeNotMatchError = _s2("BatchInferenceSizeNotMatchError")
ImageDownloadError = _s2("ImageDownloadError")
MediaDownloadError = _s2("MediaDownloadError")
UnsupportedModalityError = _s2("UnsupportedModalityError")


# ---- API-level exceptions raised from inside the path (marqo.api.exceptions in the reference) ----
class MarqoApiError(Exception):
    """stand-alone base of the API-level errors (no host application): message / code / status_code as the host's classes carry them"""
    code = "unhandled_error"
    status_code = 500

    def __init__(self, message: Optional[str] = None) -> None:
        self.message = message
        super().__init__(message)


def _api(name: str, code: str, status_code: int, parent: Optional[type] = None) -> type:
    host = _host_class(_HOST_API, name)
    return host or type(name, (parent or MarqoApiError,), {"__module__": __name__, "code": code, "status_code": status_code})


InternalError = _api("InternalError", "internal", 500)
ModelCacheManagementError = _api("ModelCacheManagementError", "model_cache_management_error", 409)
ConfigurationError = _api("ConfigurationError", "server_configuration_error", 500, parent=InternalError)
EnvVarError = _api("EnvVarError", "env_var_error", 500)   # (bad MARQO_INFERENCE_CACHE_* settings: marqo_inference_cache.py:31-52)
