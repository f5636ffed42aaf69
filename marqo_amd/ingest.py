"""Caller-side batching for bulk ingest: the step BEFORE vectorise() in add_documents.

The reference vectorises per document and per field by default (BatchingMode PER_DOCUMENT,
src/marqo/core/vespa_index/add_documents_handler.py:264-290 + tensor_fields_container.py:179-223): N = 1..10 items per
`vectorise` call, which starves any GPU.  `BulkVectoriser` is the PER_BATCH / cross-request form: callers `add()` chunks
(text or image) tagged with an opaque key as they are produced by the chunkers / download threads, and `flush()` issues
ONE `vectorise` call per (model, modality) for everything queued — the engine loaders then micro-batch on the device —
and hands the embeddings back per key, in insertion order.  With `torch.distributed` initialised (one process per GPU)
`flush()` shards the queue contiguously across ranks and all-gathers the embedding shards (the only collective on the
path, marqo_amd.parallel), so every rank returns the full result.
"""
from __future__ import annotations

import threading
from typing import Any, Dict, Hashable, List, Optional, Tuple

import numpy as np

from marqo_amd.parallel import shard_bounds
from marqo_amd.s2_inference.enums import Modality


class BulkVectoriser:
    def __init__(self, model_name: str, device: str, model_properties: Optional[dict] = None, normalize_embeddings: bool = True,
                 max_pending: int = 0, vectorise_fn=None):
        """max_pending > 0: `add()` flushes automatically once that many items are queued (bounded memory)."""
        if vectorise_fn is None:
            from marqo_amd.s2_inference.s2_inference import vectorise_ndarray as vectorise_fn
        self._vectorise = vectorise_fn
        self.model_name, self.device, self.model_properties = model_name, device, model_properties
        self.normalize = normalize_embeddings
        self.max_pending = max_pending
        self._lock = threading.Lock()
        self._pending: Dict[Modality, List[Tuple[Hashable, Any]]] = {Modality.TEXT: [], Modality.IMAGE: []}
        self._done: Dict[Hashable, np.ndarray] = {}

    def add(self, key: Hashable, content: Any, modality: Modality = Modality.TEXT) -> None:
        if modality not in self._pending:
            raise ValueError(f"unsupported modality {modality}")
        with self._lock:
            self._pending[modality].append((key, content))
            n = sum(len(v) for v in self._pending.values())
        if self.max_pending and n >= self.max_pending:
            self._run_pending()  # results stay in the store until the caller's flush()

    def pending(self) -> int:
        with self._lock:
            return sum(len(v) for v in self._pending.values())

    def _encode(self, contents: List[Any], modality: Modality) -> np.ndarray:
        """one vectorise call for the local shard (+ all-gather when running one process per GPU)"""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if world == 1:
            return self._vectorise(self.model_name, contents, model_properties=self.model_properties, device=self.device,
                                   normalize_embeddings=self.normalize, modality=modality)
        from marqo_amd.parallel import gather_embeddings
        bounds = shard_bounds(len(contents), world)
        s, e = bounds[dist.get_rank()]
        local = self._vectorise(self.model_name, contents[s:e], model_properties=self.model_properties, device=self.device,
                                normalize_embeddings=self.normalize, modality=modality) if e > s else None
        dim = torch.tensor([0 if local is None else local.shape[1]], device=self.device if self.device.startswith("cuda") else "cpu")
        dist.all_reduce(dim, op=dist.ReduceOp.MAX)
        D = int(dim.item())
        t = torch.zeros(e - s, D, dtype=torch.float32) if local is None else torch.from_numpy(np.ascontiguousarray(local, dtype=np.float32))
        t = t.to(self.device if self.device.startswith("cuda") else "cpu")
        return gather_embeddings(t, counts=[b - a for a, b in bounds]).cpu().numpy()

    def _run_pending(self) -> None:
        with self._lock:
            work = {m: v for m, v in self._pending.items() if v}
            self._pending = {Modality.TEXT: [], Modality.IMAGE: []}
        for modality, items in work.items():
            emb = self._encode([c for _, c in items], modality)
            if emb.shape[0] != len(items):
                raise RuntimeError(f"vectorise returned {emb.shape[0]} embeddings for {len(items)} items")
            with self._lock:
                for (key, _), row in zip(items, emb):
                    self._done[key] = row

    def flush(self) -> Dict[Hashable, np.ndarray]:
        """Vectorise everything still queued; returns {key: float32 [D]} for every key added since the last flush()."""
        self._run_pending()
        with self._lock:
            out, self._done = self._done, {}
        return out
