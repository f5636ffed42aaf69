"""Device-side weighted combination of sub-embeddings.

The reference combines on the host, one document / query at a time, in numpy float64:
  * multimodal-combination fields: MultiModalTensorFieldContent.tensor_field_embeddings
    (src/marqo/core/inference/tensor_fields_container.py:346-365) — mean of weight * sub-embedding, then / L2 norm;
  * weighted multi-term queries and context vectors: get_query_vectors_from_jobs
    (src/marqo/tensor_search/tensor_search.py:1940-1963) — same mean, normalised only when the norm is > 0.
For bulk ingest (thousands of multimodal documents per flush) the sub-embeddings are already in HBM after `vectorise`;
`combine_weighted` reduces all groups in ONE kernel launch (mq_weighted_combine, fp64 accumulation like the reference)
so only the combined [n_groups, D] matrix crosses PCIe.  No CPU fallback: a cuda device is required.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from marqo_amd import _lib as L

RAW, NORMALIZE, NORMALIZE_IF_NONZERO = L.MQ_COMBINE_RAW, L.MQ_COMBINE_NORMALIZE, L.MQ_COMBINE_NORMALIZE_IF_NONZERO


def combine_weighted(embeddings: Union[np.ndarray, torch.Tensor], groups: Sequence[Sequence[Tuple[int, float]]], mode: int,
                     device: str = "cuda") -> torch.Tensor:
    """embeddings fp32 [n, D] (ndarray, or a tensor already on `device`); groups[g] = [(row, weight), ...].
    Returns the combined fp32 [len(groups), D] tensor on `device` (async on the current stream)."""
    dev = torch.device(device)
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise L.MarqoHipUnavailableError(f"combine_weighted needs a cuda device (got {device!r}); there is no CPU fallback")
    lib = L.load()
    if isinstance(embeddings, np.ndarray):
        embeddings = torch.from_numpy(np.ascontiguousarray(embeddings, dtype=np.float32))
    emb = embeddings.to(device=dev, dtype=torch.float32).contiguous()
    if emb.ndim != 2:
        raise ValueError(f"embeddings must be [n, D], got {tuple(emb.shape)}")
    n, D = emb.shape
    cu = np.zeros(len(groups) + 1, dtype=np.int32)
    rows: List[int] = []
    weights: List[float] = []
    for g, terms in enumerate(groups):
        if len(terms) == 0:
            raise ValueError(f"group {g} has no terms")  # np.mean of an empty list is an error in the reference too
        for r, w in terms:
            if not 0 <= int(r) < n:
                raise IndexError(f"group {g}: row {r} out of range [0, {n})")
            rows.append(int(r))
            weights.append(float(w))
        cu[g + 1] = len(rows)
    out = torch.empty(len(groups), D, dtype=torch.float32, device=dev)
    if len(groups) == 0:
        return out
    host = torch.from_numpy(np.concatenate([cu, np.asarray(rows, dtype=np.int32), np.asarray(weights, dtype=np.float32).view(np.int32)]))
    meta = host.pin_memory().to(dev, non_blocking=True) if host.numel() > 4096 else host.to(dev)
    d_cu, d_rows, d_w = meta[:len(cu)], meta[len(cu):len(cu) + len(rows)], meta[len(cu) + len(rows):]
    with torch.cuda.device(dev):
        L.check(lib.mq_weighted_combine(emb.data_ptr(), emb.stride(0), d_rows.data_ptr(), d_w.data_ptr(), d_cu.data_ptr(), len(groups), D,
                                        mode, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "mq_weighted_combine")
    return out


def combine_multimodal_fields(sub_embeddings: Sequence[Dict[str, Sequence[float]]], weights: Dict[str, float], normalize: bool,
                              device: str = "cuda") -> np.ndarray:
    """Batch form of tensor_fields_container.py:346-365: one dict {subfield: embedding} per document, a shared weights dict;
    subfields missing from a document are skipped (the `if subfield in self.subfields` filter), order follows `weights`."""
    flat: List[Sequence[float]] = []
    groups: List[List[Tuple[int, float]]] = []
    for doc in sub_embeddings:
        terms = []
        for name, w in weights.items():
            if name in doc:
                terms.append((len(flat), w))
                flat.append(doc[name])
        groups.append(terms)
    if not flat:
        return np.zeros((0, 0), dtype=np.float32)
    emb = np.asarray(flat, dtype=np.float32)
    return combine_weighted(emb, groups, NORMALIZE if normalize else RAW, device).cpu().numpy()


def combine_query_vectors(queries: Sequence[Sequence[Tuple[Sequence[float], float]]], normalize: bool, device: str = "cuda") -> np.ndarray:
    """Batch form of tensor_search.py:1940-1963: queries[q] = [(vector, weight), ...] (vectorised terms + context tensors)."""
    flat, groups = [], []
    for terms in queries:
        g = []
        for vec, w in terms:
            g.append((len(flat), w))
            flat.append(vec)
        groups.append(g)
    if not flat:
        return np.zeros((0, 0), dtype=np.float32)
    dims = {len(v) for v in flat}
    if len(dims) != 1:
        raise ValueError(f"vectors of different dimension cannot be combined: {sorted(dims)}")
    return combine_weighted(np.asarray(flat, dtype=np.float32), groups, NORMALIZE_IF_NONZERO if normalize else RAW, device).cpu().numpy()
