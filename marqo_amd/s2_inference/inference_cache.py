"""Inference cache for string content (search / embed path).

Semantics of the reference's MarqoInferenceCache (src/marqo/inference/inference_cache/marqo_inference_cache.py:10-103,
marqo_lru_cache.py, marqo_lfu_cache.py): key = f"{model_cache_key}||{content}", value = List[float]; size 0 disables it;
LRU or LFU eviction; thread-safe.  cachetools / readerwriterlock are not in this image, so both policies are implemented
here over a plain lock (reads mutate recency/frequency, so a reader-writer lock buys nothing).
"""
from __future__ import annotations

import threading
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple, Union


from marqo_amd.s2_inference.errors import EnvVarError  # noqa: E402,F401  (the host's marqo.api.exceptions.EnvVarError when there is a host)


class _LRU:
    def __init__(self, maxsize: int):
        self.maxsize = int(maxsize)
        self._d: "OrderedDict[str, object]" = OrderedDict()
        self._lock = threading.Lock()

    def get(self, key, default=None):
        with self._lock:
            if key not in self._d:
                return default
            self._d.move_to_end(key)
            return self._d[key]

    def __setitem__(self, key, value):
        with self._lock:
            if key in self._d:
                self._d.move_to_end(key)
            self._d[key] = value
            while len(self._d) > self.maxsize:
                self._d.popitem(last=False)

    def __getitem__(self, key):
        with self._lock:
            self._d.move_to_end(key)
            return self._d[key]

    def set(self, key, value) -> None:
        """(the reference's classes expose both spellings: marqo_lru_cache.py:24-26)"""
        self[key] = value

    def popitem(self) -> None:
        """evict the entry the policy would evict next (marqo_lru_cache.py:43-45)"""
        with self._lock:
            if self._d:
                self._pop_victim()

    def _pop_victim(self) -> None:
        self._d.popitem(last=False)

    def __contains__(self, key) -> bool:
        with self._lock:
            return key in self._d

    def __len__(self) -> int:
        return len(self._d)

    def clear(self) -> None:
        with self._lock:
            self._d.clear()

    @property
    def currsize(self) -> int:
        return len(self._d)


class _LFU(_LRU):
    """least-frequently-used; ties evict the least recently used of the minimum-frequency entries"""

    def __init__(self, maxsize: int):
        super().__init__(maxsize)
        self._freq: Dict[str, int] = {}

    def get(self, key, default=None):
        with self._lock:
            if key not in self._d:
                return default
            self._freq[key] += 1
            self._d.move_to_end(key)
            return self._d[key]

    def __getitem__(self, key):
        with self._lock:
            self._freq[key] += 1
            self._d.move_to_end(key)
            return self._d[key]

    def __setitem__(self, key, value):
        with self._lock:
            if key in self._d:
                self._d[key] = value
                self._freq[key] += 1
                self._d.move_to_end(key)
                return
            while len(self._d) >= self.maxsize and self._d:
                self._pop_victim()
            self._d[key] = value
            self._freq[key] = 1

    def _pop_victim(self) -> None:
        victim = min(self._d, key=lambda k: self._freq[k])  # OrderedDict iteration = recency order -> LRU tie-break
        del self._d[victim]
        del self._freq[victim]

    def clear(self) -> None:
        with self._lock:
            self._d.clear()
            self._freq.clear()


# the reference's class names (inference/inference_cache/marqo_lru_cache.py, marqo_lfu_cache.py): same methods — get / set / [] / in / len /
# popitem / clear / maxsize / currsize
MarqoLRUCache = _LRU
MarqoLFUCache = _LFU


class MarqoInferenceCache:
    _CACHE_TYPES_MAPPING = {"LRU": _LRU, "LFU": _LFU}

    def __init__(self, cache_size: int = 0, cache_type: Union[None, str] = "LRU"):
        self._cache = self._build_cache(cache_size, cache_type)

    def _build_cache(self, cache_size, cache_type):
        if not isinstance(cache_size, int) or cache_size < 0:
            raise EnvVarError(f"Invalid cache size: {cache_size}. Must be a non-negative integer. Please set the "
                              f"'MARQO_INFERENCE_CACHE_SIZE' environment variable to a non-negative integer.")
        if cache_size == 0:
            return None
        cache_type = getattr(cache_type, "value", cache_type)
        if cache_type not in self._CACHE_TYPES_MAPPING:
            raise EnvVarError(f"Invalid cache type: {cache_type}. Must be one of {list(self._CACHE_TYPES_MAPPING)}. Please set the "
                              f"'MARQO_INFERENCE_CACHE_TYPE' environment variable to one of the valid cache types.")
        return self._CACHE_TYPES_MAPPING[cache_type](maxsize=cache_size)

    @staticmethod
    def _generate_key(model_cache_key: str, content: str) -> str:
        if not isinstance(model_cache_key, str):
            raise TypeError(f"model_cache_key must be a string, not {type(model_cache_key)}")
        if not isinstance(content, str):
            raise TypeError(f"content must be a string, not {type(content)}")
        return f"{model_cache_key}||{content}"

    def get(self, model_cache_key: str, content: str, default=None) -> Optional[List[float]]:
        return self._cache.get(self._generate_key(model_cache_key, content), default)

    def set(self, model_cache_key: str, content: str, value: List[float]) -> None:
        self._cache[self._generate_key(model_cache_key, content)] = value

    def __contains__(self, item: Tuple) -> bool:
        if len(item) != 2:
            raise ValueError("MarqoInferenceCache received an unsupported input for 'in' operation. Expected input is a tuple "
                             "with 'model-cache-key' and 'content'. E.g., ('my-model-cache-key', 'content'). ")
        return self._generate_key(*item) in self._cache

    def clear(self) -> None:
        if self._cache is not None:
            self._cache.clear()

    def is_enabled(self) -> bool:
        return self._cache is not None

    @property
    def maxsize(self) -> int:
        return self._cache.maxsize

    @property
    def currsize(self) -> int:
        return self._cache.currsize
