"""Error hierarchy of the vectorise() path.

Same class names and inheritance as the reference (src/marqo/s2_inference/errors.py:4-72) so callers'
`except` clauses (tensor_fields_container.py:155-163, tensor_search.py:1899-1911) keep working, plus
the three API-level exceptions the path raises (src/marqo/api/exceptions.py: InternalError,
ModelCacheManagementError, ConfigurationError).
"""
from typing import Optional


class S2InferenceError(Exception):
    def __init__(self, message: Optional[str] = None) -> None:
        self.message = message
        super().__init__(self.message)


class MediaMismatchError(S2InferenceError): pass
class ChunkerError(S2InferenceError): pass
class ChunkerMethodProcessError(S2InferenceError): pass
class VectoriseError(S2InferenceError): pass
class InvalidModelPropertiesError(S2InferenceError): pass
class UnknownModelError(S2InferenceError): pass
class ModelLoadError(S2InferenceError): pass
class ModelDownloadError(S2InferenceError): pass
class ModelNotInCacheError(S2InferenceError): pass
class IncompatibleModelDeviceError(S2InferenceError): pass
class BatchInferenceSizeNotMatchError(S2InferenceError): pass
class ImageDownloadError(S2InferenceError): pass
class MediaDownloadError(S2InferenceError): pass
class UnsupportedModalityError(S2InferenceError): pass


# ---- API-level exceptions raised from inside the path (marqo.api.exceptions in the reference) ----
class MarqoApiError(Exception):
    code = "unhandled_error"
    status_code = 500

    def __init__(self, message: Optional[str] = None) -> None:
        self.message = message
        super().__init__(message)


class InternalError(MarqoApiError):
    code = "internal"
    status_code = 500


class ModelCacheManagementError(MarqoApiError):
    code = "model_cache_management_error"
    status_code = 409


class ConfigurationError(InternalError):
    code = "configuration_error"
    status_code = 500
