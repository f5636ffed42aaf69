"""Error hierarchy of the vectorise() path.

Same class names and inheritance as the reference (src/marqo/s2_inference/errors.py:4-72), plus the three API-level exceptions the path
raises (src/marqo/api/exceptions.py:128-130,216-219,246-248: ModelCacheManagementError, InternalError, ConfigurationError).

Names alone do not make the callers' `except` clauses work: the reference catches ITS OWN classes
(`except (s2_inference_errors.UnknownModelError, ...)` in core/inference/tensor_fields_container.py:155-163 and
tensor_search/tensor_search.py:1899-1911; the FastAPI handlers dispatch on `marqo.api.exceptions.MarqoWebError`).  So when this engine is
installed INTO the host application — the package `marqo` is importable — every class below also derives from the host's class of the same
name: an `UnknownModelError` raised here IS a `marqo.s2_inference.errors.UnknownModelError`, an `InternalError` IS a
`marqo.api.exceptions.InternalError` (message suffix, status code and error code as the host defines them), and swapping the
`s2_inference` import is the whole integration (tests/test_ref_parity.py runs the reference's own Vectoriser classes on top of this
module to check exactly that).  Without a host (`MARQO_AMD_HOST_ERRORS=0`, or no `marqo` package: the GPU box, the tests) they are
plain classes with the reference's constructor, attributes and codes.
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


def _host_class(module, name: str):
    c = getattr(module, name, None) if module is not None else None
    return c if isinstance(c, type) and issubclass(c, Exception) else None


_HostS2Base = _host_class(_HOST_S2, "S2InferenceError")


class S2InferenceError(_HostS2Base or Exception):
    def __init__(self, message: Optional[str] = None) -> None:
        self.message = message
        Exception.__init__(self, self.message)


def _s2(name: str) -> type:
    """our subclass of S2InferenceError called `name`, also a subclass of the host's class of that name when there is a host"""
    host = _host_class(_HOST_S2, name)
    bases = (S2InferenceError, host) if host is not None and _HostS2Base is not None and issubclass(host, _HostS2Base) else (S2InferenceError,)
    return type(name, bases, {"__module__": __name__, "__doc__": f"src/marqo/s2_inference/errors.py: {name}"})


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
    """stand-alone base of the three API-level errors (no host application): message / code / status_code as the host's classes carry them"""
    code = "unhandled_error"
    status_code = 500

    def __init__(self, message: Optional[str] = None) -> None:
        self.message = message
        super().__init__(message)


def _api(name: str, code: str, status_code: int, parent: Optional[type] = None) -> type:
    """With a host: a subclass of the host's class (its __init__, message suffix, code, status code — whatever the host's handlers expect)
    and of `parent` (our class one level up, so `except InternalError` inside this package still sees a ConfigurationError).
    Stand-alone: a MarqoApiError subclass with the reference's code / status."""
    host = _host_class(_HOST_API, name)
    if host is not None:
        try:
            return type(name, (parent, host) if parent is not None else (host,), {"__module__": __name__})
        except TypeError:   # inconsistent MRO in an unexpected host hierarchy: the host's class alone
            return type(name, (host,), {"__module__": __name__})
    return type(name, (parent or MarqoApiError,), {"__module__": __name__, "code": code, "status_code": status_code})


InternalError = _api("InternalError", "internal", 500)
ModelCacheManagementError = _api("ModelCacheManagementError", "model_cache_management_error", 409)
ConfigurationError = _api("ConfigurationError", "server_configuration_error", 500, parent=InternalError)


# marqo.api.exceptions.EnvVarError (api/exceptions.py:24-28: a MarqoError whose constructor only stores the message), raised by the
# inference cache for bad MARQO_INFERENCE_CACHE_* settings (inference/inference_cache/marqo_inference_cache.py:31-52)
_HostEnvVarError = _host_class(_HOST_API, "EnvVarError")


class EnvVarError(_HostEnvVarError or Exception):
    code = "env_var_error"

    def __init__(self, message: Optional[str] = None) -> None:
        self.message = message
        Exception.__init__(self, message)
