"""ORACLE — test infrastructure only.  Never imported by the product (marqo_amd/*).

numpy (float64) restatement of the two places where the reference combines sub-embeddings with weights:

* ``combine_multimodal``  <- MultiModalTensorFieldContent.tensor_field_embeddings
                             (src/marqo/core/inference/tensor_fields_container.py:346-365):
                             ``np.squeeze(np.mean([np.array(e) * w ...], axis=0))`` then ``/ np.linalg.norm`` when
                             normalize_embeddings (no zero guard: a zero vector gives NaN).
* ``combine_query``       <- get_query_vectors_from_jobs (src/marqo/tensor_search/tensor_search.py:1940-1963):
                             ``np.mean([np.asarray(vec) * weight ...], axis=0)`` then divided by the norm only if norm > 0.

Both operate on Python float lists in the reference, i.e. float64; parity of the fp32-output device kernel is therefore
asserted to fp32 rounding (rtol 1e-6).  Pinned by tests/test_combine.py against literal numpy expressions.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np


def combine_multimodal(vectors: Sequence[Sequence[float]], weights: Sequence[float], normalize: bool) -> np.ndarray:
    combo = [np.array(v, dtype=np.float64) * w for v, w in zip(vectors, weights)]
    chunk = np.squeeze(np.mean(combo, axis=0))
    if normalize:
        with np.errstate(invalid="ignore", divide="ignore"):
            chunk = chunk / np.linalg.norm(chunk)
    return chunk


def combine_query(vectors: Sequence[Sequence[float]], weights: Sequence[float], normalize: bool) -> np.ndarray:
    weighted = [np.asarray(v, dtype=np.float64) * w for v, w in zip(vectors, weights)]
    merged = np.mean(weighted, axis=0)
    if normalize:
        norm = np.linalg.norm(merged, axis=-1, keepdims=True)
        if norm > 0:
            merged /= np.linalg.norm(merged, axis=-1, keepdims=True)
    return merged
