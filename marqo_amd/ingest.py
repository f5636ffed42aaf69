"""Caller-side batching for bulk ingest: the step BEFORE vectorise() in add_documents.

The reference vectorises per document and per field by default (BatchingMode PER_DOCUMENT,
src/marqo/core/vespa_index/add_documents_handler.py:264-290 + tensor_fields_container.py:179-223): N = 1..10 items per
`vectorise` call, which starves any GPU.  `BulkVectoriser` is the PER_BATCH / cross-request form: callers `add()` chunks
(text or image) tagged with an opaque key as they are produced by the chunkers / download threads, and `flush()` issues
ONE `vectorise` call per (model, modality) for everything queued — the engine loaders then micro-batch on the device —
and hands the embeddings back per key, in insertion order.

With `torch.distributed` initialised (one process per GPU) `flush()` shards each modality's queue across the ranks — images
contiguously (equal cost after the resize), texts balanced by estimated token count (marqo_amd.parallel.balanced_shards) — every
rank encodes its shard and keeps it IN HBM (`vectorise_device`), and ONE all_gather per modality (RCCL over xGMI) rebuilds the full
[N, D] matrix in request order on every rank; the embedding width comes from the model properties, nothing else is exchanged.
"""
from __future__ import annotations

import os
import threading
import time
from typing import Any, Callable, Dict, Hashable, List, Optional, Tuple

import numpy as np

from marqo_amd.parallel import agree_on_shard, balanced_shards, contiguous_shards, gather_embeddings, gather_embeddings_to_root
from marqo_amd.s2_inference.enums import Modality


# a flush with BOTH modalities on a GPU prepares them on two host threads (A/B knob): the text side (tokeniser, launch sequence) runs on a helper
# thread while the caller's thread exports / packs the images — the native stager and the launches release the GIL
PARALLEL_MODALITIES = os.environ.get("MARQO_AMD_INGEST_THREADS", "1") != "0"


_now = time.perf_counter


def estimate_tokens(text: Any) -> float:
    """cheap per-item cost for balancing text shards: ~ BPE / WordPiece tokens of a string (4 characters per token + the two
    specials); anything that is not a string costs 1"""
    return 2.0 + len(text) / 4.0 if isinstance(text, str) else 1.0


class PeerShardError(RuntimeError):
    """raised on the ranks whose own shard encoded fine when ANOTHER rank's shard of the same flush failed (see BulkVectoriser._encode)"""


class _DeferredRows:
    """[n, D] rows a tower has been ENQUEUED for.  Device rows start their copy into pinned host memory right away, on the caller's stream —
    which, at this point, waits for exactly the work these rows depend on — and `numpy()` only waits for that copy's event.  (A blocking
    `.cpu()` issued later would queue behind everything the thread has enqueued SINCE — with a request in flight: behind the next request's
    towers, which is what kept RequestShardedIngest's pipeline from overlapping anything, profiles/r04s_stream_timeline.txt.)"""

    def __init__(self, rows):
        import torch
        self.rows, self.host, self.event = rows, None, None
        if isinstance(rows, torch.Tensor) and rows.is_cuda:
            self.host = torch.empty(rows.shape, dtype=rows.dtype, pin_memory=True)
            self.host.copy_(rows, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record(torch.cuda.current_stream(rows.device))
            # `rows` stays referenced until the copy has completed: its block belongs to the tower's stream, the copy runs on this one

    @property
    def shape(self):
        return self.rows.shape

    def numpy(self) -> np.ndarray:
        import torch
        if self.event is not None:
            self.event.synchronize()
            out = self.host.numpy().copy()     # (the pinned block goes back to the host allocator; the rows live on for as long as the caller keeps them)
            self.rows = self.host = self.event = None
            return out
        return self.rows.cpu().numpy() if isinstance(self.rows, torch.Tensor) else self.rows


    def numpy_into(self, dst: np.ndarray) -> None:
        """the rows straight into `dst` ([n, D] float32, e.g. a slice of RequestShardedIngest's row slab): ONE host copy, no intermediate array"""
        import torch
        if self.event is not None:
            self.event.synchronize()
            np.copyto(dst, self.host.numpy())
            self.rows = self.host = self.event = None
            return
        np.copyto(dst, self.rows.cpu().numpy() if isinstance(self.rows, torch.Tensor) else self.rows)


class _DeviceRows:
    """[n, D] rows a tower has been enqueued for that STAY in HBM (RequestShardedIngest on RCCL ranks: the rows meet their first host at the root,
    behind the gather).  `tensor()` waits for the tower (an asynchronous device fault surfaces there, like at _DeferredRows' copy) and hands the
    device tensor over."""

    def __init__(self, rows):
        import torch
        self.rows = rows
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(rows.device))

    @property
    def shape(self):
        return self.rows.shape

    def tensor(self):
        self.event.synchronize()
        return self.rows

    def numpy(self) -> np.ndarray:
        return self.tensor().cpu().numpy()

    def numpy_into(self, dst: np.ndarray) -> None:
        np.copyto(dst, self.numpy())


class _RowSlab:
    """a rank's filed rows on the host: a short list of large [chunk, D] float32 arrays.  Every settled group's rows are copied ONCE, from the pinned
    block their D2H landed in, to the current chunk's tail (a group never straddles two chunks); a full chunk is followed by a fresh one — nothing is
    ever re-copied while the stream runs.  Row positions count the rows in filing order, i.e. positions in the concatenation of the chunks' used
    parts: collect() slices rows out of the chunks (single rank: no copy at all) or concatenates them once for the gather."""
    CHUNK_ROWS = 65536

    def __init__(self):
        self.chunks: List[np.ndarray] = []
        self.used: List[int] = []
        self.n = 0

    def tail(self, n: int, D: int) -> np.ndarray:
        """a writable view of the next n rows (commit() makes them part of the slab)"""
        if self.chunks and self.chunks[-1].shape[1] != D:
            raise ValueError(f"row width changed inside one stream: {self.chunks[-1].shape[1]} -> {D}")
        if not self.chunks or self.used[-1] + n > self.chunks[-1].shape[0]:
            self.chunks.append(np.empty((max(self.CHUNK_ROWS, n), D), dtype=np.float32))
            self.used.append(0)
        return self.chunks[-1][self.used[-1]:self.used[-1] + n]

    def commit(self, n: int) -> int:
        base = self.n
        self.used[-1] += n
        self.n += n
        return base

    def take(self) -> List[np.ndarray]:
        """the used parts of the chunks, in filing order; the slab is empty afterwards"""
        out = [c[:u] for c, u in zip(self.chunks, self.used) if u]
        self.chunks, self.used, self.n = [], [], 0
        return out


def _rows_at(parts: List[np.ndarray], positions: np.ndarray):
    """(chunk index, row inside the chunk) per position of the concatenation of `parts`, vectorised"""
    starts = np.zeros(len(parts) + 1, dtype=np.int64)
    np.cumsum([p.shape[0] for p in parts], out=starts[1:])
    which = np.searchsorted(starts, positions, side="right") - 1
    return which, positions - starts[which]


class BulkVectoriser:
    def __init__(self, model_name: str, device: str, model_properties: Optional[dict] = None, normalize_embeddings: bool = True,
                 max_pending: int = 0, vectorise_fn: Optional[Callable] = None):
        """max_pending > 0: `add()` flushes automatically once that many items are queued (bounded memory).
        vectorise_fn(model_name, contents, model_properties=, device=, normalize_embeddings=, modality=) -> [n, D] ndarray or
        torch.Tensor; default: `vectorise_ndarray`, or `vectorise_device` when sharding over a cuda process group."""
        self._vectorise = vectorise_fn
        self.model_name, self.device, self.model_properties = model_name, device, model_properties
        self.normalize = normalize_embeddings
        self.max_pending = max_pending
        self._lock = threading.Lock()
        self._pending: Dict[Modality, List[Tuple[Hashable, Any]]] = {Modality.TEXT: [], Modality.IMAGE: []}
        self._done: Dict[Hashable, np.ndarray] = {}
        self.force_collective = False   # tests: take the sharded path (and run the collective) even in a 1-rank process group
        self.local_only = False         # RequestShardedIngest: requests are owned by ONE rank, nothing is sharded inside a request
        self.rows_on_device = False     # RequestShardedIngest on RCCL ranks: a flush's rows stay in HBM (_DeviceRows) instead of starting their D2H
        self._helper = None             # second host thread of a two-modality flush (created on first use)
        self.two_threads = True         # (False: both modalities on the caller's thread; measurement knob)
        self._model_loaded = False      # a vectorise call of this object has returned (the model is in the cache)

    def add(self, key: Hashable, content: Any, modality: Modality = Modality.TEXT) -> None:
        if modality not in self._pending:
            raise ValueError(f"unsupported modality {modality}")
        with self._lock:
            self._pending[modality].append((key, content))
            n = sum(len(v) for v in self._pending.values())
        if self.max_pending and n >= self.max_pending:
            self._run_pending()  # results stay in the store until the caller's flush()

    def add_many(self, pairs: List[Tuple[Hashable, Any]], modality: Modality = Modality.TEXT) -> None:
        """`add()` for a list of (key, content) of one modality under one lock acquisition (no automatic max_pending flush in between)"""
        if modality not in self._pending:
            raise ValueError(f"unsupported modality {modality}")
        if not pairs:
            return
        with self._lock:
            self._pending[modality].extend(pairs)
            n = sum(len(v) for v in self._pending.values())
        if self.max_pending and n >= self.max_pending:
            self._run_pending()

    def pending(self) -> int:
        with self._lock:
            return sum(len(v) for v in self._pending.values())

    def _call(self, fn, contents, modality):
        return fn(self.model_name, contents, model_properties=self.model_properties, device=self.device,
                  normalize_embeddings=self.normalize, modality=modality)

    def _dimensions(self) -> Optional[int]:
        props = self.model_properties
        if props is None:
            try:
                from marqo_amd.s2_inference.s2_inference import get_model_properties_from_registry
                props = get_model_properties_from_registry(self.model_name)
            except Exception:  # noqa: BLE001 - unknown to the registry: the width is taken from the local shard below
                props = None
        d = (props or {}).get("dimensions")
        return int(d) if isinstance(d, int) and d > 0 else None

    def _encode(self, contents: List[Any], modality: Modality, defer: bool = False):
        """one vectorise call for the local shard (+ ONE all-gather when running one process per GPU) -> float32 [n, D] ndarray.
        defer=True (single process on a GPU): the rows may come back as a DEVICE tensor whose kernels are only enqueued — the caller
        copies it to the host after it has staged the next modality, so that modality's host work overlaps this one's GPU work."""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() and not self.local_only else 1
        if world == 1 and not (self.force_collective and dist.is_available() and dist.is_initialized()):
            fn = self._vectorise
            if fn is None:
                if defer and str(self.device).startswith("cuda") and torch.cuda.is_available():
                    from marqo_amd.s2_inference.s2_inference import vectorise_device
                    return self._call(vectorise_device, contents, modality)
                from marqo_amd.s2_inference.s2_inference import vectorise_ndarray as fn
            out = self._call(fn, contents, modality)
            if defer and isinstance(out, torch.Tensor):
                return out                       # (a vectorise_fn that hands back tensors: copied by the caller, like vectorise_device's)
            return out.cpu().numpy() if isinstance(out, torch.Tensor) else out
        on_gpu = dist.get_backend() == "nccl" and str(self.device).startswith("cuda")
        fn = self._vectorise
        if fn is None:
            if on_gpu:
                from marqo_amd.s2_inference.s2_inference import vectorise_device as fn
            else:
                from marqo_amd.s2_inference.s2_inference import vectorise_ndarray as fn
        plan = (balanced_shards([estimate_tokens(c) for c in contents], world) if modality == Modality.TEXT
                else contiguous_shards(len(contents), world))
        mine = plan.items[dist.get_rank()]
        where = self.device if on_gpu else "cpu"
        # The local encode may raise (one undecodable image in this rank's shard).  Leaving now would strand the peers in the all_gather
        # until the RCCL timeout, so failure is made collective: ONE 2-int all_reduce carries (everyone ok?, embedding width) in front of
        # the data collective; when any rank failed, EVERY rank raises after it (the failing rank its own exception, the others a
        # PeerShardError) and every rank re-queues — the re-queue contract of _run_pending holds on all ranks alike.
        local, failure = None, None
        try:
            local = self._call(fn, [contents[i] for i in mine], modality) if mine else None
            if local is not None and not isinstance(local, torch.Tensor):
                local = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float32))
            if local is not None and local.shape[0] != len(mine):
                raise RuntimeError(f"vectorise returned {local.shape[0]} embeddings for {len(mine)} items")
        except BaseException as e:  # noqa: BLE001 - re-raised below, after the agreement
            failure, local = e, None
        # the width comes from the rows actually produced (a loaded model knows its width; the registry's 'dimensions' may not match a
        # custom checkpoint), agreed over the ranks so that a rank without items sizes its empty shard like everyone else
        all_ok, D = agree_on_shard(failure is None, local.shape[1] if local is not None else 0, where)
        if not all_ok:
            raise failure if failure is not None else PeerShardError(
                f"BulkVectoriser: another rank failed to encode its shard of this {modality} flush; nothing was gathered")
        if D <= 0:
            D = self._dimensions() or 0
        if D <= 0:
            raise RuntimeError("BulkVectoriser: no rank received items and the model properties carry no 'dimensions'")
        local = torch.zeros(0, D, dtype=torch.float32, device=where) if local is None else local.to(device=where, dtype=torch.float32)
        full = plan.restore(gather_embeddings(local, counts=plan.counts, force_collective=self.force_collective))
        return full.cpu().numpy()

    def _enqueue_one(self, modality: Modality, items):
        """one modality's vectorise call -> ((modality, items, rows: ndarray | _DeferredRows) or None, exception or None); a failure puts the
        items back at the front of the queue"""
        try:
            emb = self._encode([c for _, c in items], modality, defer=True)
            self._model_loaded = True
            if emb.shape[0] != len(items):
                raise RuntimeError(f"vectorise returned {emb.shape[0]} embeddings for {len(items)} items")
            if not isinstance(emb, np.ndarray):
                emb = _DeviceRows(emb) if self.rows_on_device and emb.is_cuda else _DeferredRows(emb)
            return (modality, items, emb), None
        except BaseException as e:  # noqa: BLE001 - re-raised by the caller, after the modalities that did run are stored
            with self._lock:
                self._pending[modality] = items + self._pending[modality]
            return None, e

    def _enqueue_pending(self):
        """pops ONE modality at a time and runs its vectorise call; if that raises (e.g. one undecodable image) the popped items go back to the
        front of the queue — nothing is lost, the caller sees the exception and may drop the offending key and flush again.
        On a GPU the modalities are pipelined: the first one's kernels are enqueued (rows stay in HBM, their copy to the host is enqueued behind
        them), the second one is tokenised / packed / enqueued while they run — on a second host thread when the engine's own loaders are
        used (`_two_threads`).  -> ([(modality, items, rows)], first exception or None)"""
        if self._two_threads():
            with self._lock:
                texts, self._pending[Modality.TEXT] = self._pending[Modality.TEXT], []
                images, self._pending[Modality.IMAGE] = self._pending[Modality.IMAGE], []
            if texts and images:
                if self._helper is None:
                    from concurrent.futures import ThreadPoolExecutor
                    self._helper = ThreadPoolExecutor(max_workers=1, thread_name_prefix="marqo-amd-ingest-text")
                fut = self._helper.submit(self._enqueue_one, Modality.TEXT, texts)
                img, img_fail = self._enqueue_one(Modality.IMAGE, images)
                txt, txt_fail = fut.result()
                return [e for e in (txt, img) if e is not None], txt_fail or img_fail
            with self._lock:   # one modality only: the sequential form below
                self._pending[Modality.TEXT] = texts + self._pending[Modality.TEXT]
                self._pending[Modality.IMAGE] = images + self._pending[Modality.IMAGE]
        enqueued, failure = [], None
        for modality in (Modality.TEXT, Modality.IMAGE):
            with self._lock:
                items, self._pending[modality] = self._pending[modality], []
            if not items:
                continue
            entry, failure = self._enqueue_one(modality, items)
            if failure is not None:
                break
            enqueued.append(entry)
        return enqueued, failure

    def _two_threads(self) -> bool:
        import torch
        import torch.distributed as dist
        if not (PARALLEL_MODALITIES and self.two_threads) or self._vectorise is not None or self.force_collective or not self._model_loaded:
            return False    # (the first flush loads the model: the reference's model cache rejects two concurrent loads, s2_inference.py:348-394)
        if not (str(self.device).startswith("cuda") and torch.cuda.is_available()):
            return False
        return self.local_only or not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)

    def _store(self, enqueued):
        """copies what `_enqueue_pending` left in HBM to the host and files the rows under their keys -> first exception or None"""
        import torch
        failure = None
        for modality, items, emb in enqueued:
            try:
                if not isinstance(emb, np.ndarray):
                    emb = emb.numpy()
            except BaseException as e:  # noqa: BLE001 - an asynchronous device error surfaces at the copy: same re-queue rule
                with self._lock:
                    self._pending[modality] = items + self._pending[modality]
                failure = failure or e
                continue
            with self._lock:
                for (key, _), row in zip(items, emb):
                    self._done[key] = row
        return failure

    def _run_pending(self) -> None:
        enqueued, failure = self._enqueue_pending()
        late = self._store(enqueued)
        failure = failure or late
        if failure is not None:
            raise failure

    def flush_async(self) -> "PendingFlush":
        """`flush()` in two halves: everything queued is tokenised / packed / ENQUEUED now (a failure there raises now, with the re-queue
        rule of flush()); the returned handle's `result()` copies the rows to the host and returns what flush() would have.  Between the two
        the GPU works on this flush while the caller prepares the next one (RequestShardedIngest keeps one request in flight that way).
        Until `result()` ran, the keys of this flush are not visible to flush() / another flush_async()."""
        enqueued, failure = self._enqueue_pending()
        if failure is not None:
            self._store(enqueued)      # the modality that did run is kept for the retry, as in flush()
            raise failure
        with self._lock:
            prior, self._done = self._done, {}
        return PendingFlush(self, enqueued, prior)

    def discard(self, key: Hashable) -> int:
        """drop every queued item with this key (e.g. the image a failed flush reported); returns how many were dropped"""
        with self._lock:
            n = 0
            for m, items in self._pending.items():
                kept = [(k, c) for k, c in items if k != key]
                n += len(items) - len(kept)
                self._pending[m] = kept
            return n

    def reset(self) -> int:
        """forget everything: queued items of both modalities AND rows already encoded but not yet handed out (a failed flush keeps the
        modality that did run in the store); returns how many items / rows were dropped"""
        with self._lock:
            n = sum(len(v) for v in self._pending.values()) + len(self._done)
            for m in self._pending:
                self._pending[m] = []
            self._done = {}
            return n

    def flush(self) -> Dict[Hashable, np.ndarray]:
        """Vectorise everything still queued; returns {key: float32 [D]} for every key added since the last flush()."""
        self._run_pending()
        with self._lock:
            out, self._done = self._done, {}
        return out


class PendingFlush:
    """handle of BulkVectoriser.flush_async(): rows still in HBM (or already on the host, for CPU / stub vectorise functions)"""

    def __init__(self, bulk: BulkVectoriser, enqueued, prior: Dict[Hashable, np.ndarray]):
        self._bulk, self._enqueued, self._rows = bulk, enqueued, prior

    def result(self) -> Dict[Hashable, np.ndarray]:
        """{key: float32 [D]} of this flush (+ rows an automatic max_pending flush had already stored); a device error that surfaces at the copy
        re-queues that modality's items in front of the owner's queue and raises"""
        import torch
        if self._enqueued is not None:
            enqueued, self._enqueued = self._enqueued, None
            failure = None
            for modality, items, emb in enqueued:
                try:
                    if not isinstance(emb, np.ndarray):
                        emb = emb.numpy()
                except BaseException as e:  # noqa: BLE001
                    with self._bulk._lock:
                        self._bulk._pending[modality] = items + self._bulk._pending[modality]
                    failure = failure or e
                    continue
                for (key, _), row in zip(items, emb):
                    self._rows[key] = row
            if failure is not None:
                raise failure
        return self._rows


    def result_blocks(self):
        """the same hand-over WITHOUT the per-key dictionary: [(modality, items, rows)] with rows = the flush's [n, D] block per modality, still
        lazy (_DeferredRows / _DeviceRows: the caller copies it where it wants it) or an ndarray.  Waits for nothing; a device error surfaces when
        the caller touches the rows — `requeue(modality, items)` then puts that modality back, as result() does."""
        if self._rows:
            raise RuntimeError("result_blocks(): rows of an automatic max_pending flush are pending; use result()")
        enqueued, self._enqueued = self._enqueued, None
        return enqueued or []

    def requeue(self, modality, items) -> None:
        with self._bulk._lock:
            self._bulk._pending[modality] = items + self._bulk._pending[modality]


# Cross-request micro-batching of the ingest stream (north_star: "documents are dynamically micro-batched").  A 128-document request is
# 6 400 image rows + ~4 300 packed text rows: every GEMM of its towers fills about half of the chip's 512 resident tile slots, so the stream ran
# at ~47 % of what the same kernels deliver at full batches (profiles/r04ad_stream_procs.txt).  Requests are therefore MERGED until one side is
# chip-filling: >= MERGE_IMAGES images or >= MERGE_TEXT_TOKENS estimated text tokens (or MERGE_MAX_REQUESTS requests), or until the oldest waiting
# request is MERGE_DEADLINE_MS old — then ONE tower call per modality runs for the whole group (text and images on two host threads = two HIP
# streams) and the rows are scattered back per request.  MARQO_AMD_INGEST_MERGE_IMAGES=0 switches merging off (one group per request).
MERGE_IMAGES = int(os.environ.get("MARQO_AMD_INGEST_MERGE_IMAGES", "512"))
MERGE_TEXT_TOKENS = int(os.environ.get("MARQO_AMD_INGEST_MERGE_TEXT_TOKENS", "24576"))
MERGE_MAX_REQUESTS = int(os.environ.get("MARQO_AMD_INGEST_MERGE_MAX_REQUESTS", "64"))
MERGE_DEADLINE_MS = float(os.environ.get("MARQO_AMD_INGEST_MERGE_DEADLINE_MS", "2"))
PIPELINE_DEPTH = int(os.environ.get("MARQO_AMD_INGEST_PIPELINE_DEPTH", "1"))     # groups in flight per rank
OVERLAP_SETTLE = os.environ.get("MARQO_AMD_INGEST_OVERLAP_SETTLE", "1") != "0"


def _deadline_loop(ref) -> None:
    """body of an ingest object's deadline thread; holds the object only while it works, so an abandoned ingest can be collected"""
    while True:
        ing = ref()
        if ing is None or ing._closed:
            return
        ing._deadline_tick()
        del ing


class _Request:
    """one submitted request, as RequestShardedIngest keeps it until its rows are filed"""
    __slots__ = ("keys", "texts", "images", "kinds")

    def __init__(self):
        self.keys: List[Tuple[int, Hashable]] = []                 # (request index, key) in submission order
        self.kinds = bytearray()                                   # 0 = text, 1 = image per key (the row's place inside its group's blocks)
        self.texts: List[Tuple[Tuple[int, Hashable], Any]] = []    # ((request index, key), content) per modality
        self.images: List[Tuple[Tuple[int, Hashable], Any]] = []


class RequestShardedIngest:
    """BASELINE configs[3] ("add_documents bulk ingest: mixed text + image docs sharded across the GPUs, RCCL gather"), sharded AT THE
    SOURCE: ranks own disjoint REQUESTS.  Request i (a batch of <= 128 documents, what one add_documents call carries,
    add_documents_handler.py:344-373) belongs to rank i % world; only its owner ever touches its documents (decodes its images, tokenises
    its texts, stages its bytes) and keeps the [n, D] rows.  On its rank, consecutive owned requests are merged into chip-filling GROUPS (see
    MERGE_* above: the reference's PER_BATCH idea, add_docs.py:325-381, taken across requests); a group runs through a local BulkVectoriser
    — one tower call per modality, the two on two host threads / HIP streams, no collective — with ONE group in flight: group g + 1 is
    tokenised / packed / enqueued while group g's kernels run.  `collect()` is the one data-path collective: a `gather` of the ranks' rows
    onto the root (the process that feeds the document store), NOT an all_gather: nothing is replicated, non-root ranks copy nothing to
    their host.  The reference has no counterpart (one device string per call, tensor_search/utils.py:90-123).

    Failure isolation: a request is all-or-nothing and never takes another one down.  When a merged group fails — on the host (an
    undecodable image) or at the deferred copy (an asynchronous device fault) — its requests are re-run ONE BY ONE, synchronously: the ones
    that encode are filed, the ones that raise are recorded (`failed`, `errors[i]`, collect()'s `failed_requests`).  submit(i) raises only
    what request i itself raised, and only when request i's group was launched by that very call.

    usage, same code on every rank:
        ing = RequestShardedIngest(model, device)
        for i, request in enumerate(stream):
            if ing.owns(i): ing.submit(i, request)          # request: [(key, content, Modality), ...]
        rows = ing.collect()                                # root: {request index: {key: float32 [D]}}, other ranks: {}
    """

    def __init__(self, model_name: str, device: str, model_properties: Optional[dict] = None, normalize_embeddings: bool = True,
                 vectorise_fn: Optional[Callable] = None, root: int = 0, merge_images: Optional[int] = None,
                 merge_text_tokens: Optional[int] = None, merge_deadline_ms: Optional[float] = None):
        import torch.distributed as dist
        self._dist = dist if dist.is_available() and dist.is_initialized() else None
        self.world = self._dist.get_world_size() if self._dist else 1
        self.rank = self._dist.get_rank() if self._dist else 0
        self.root, self.device = root, device
        self._bulk = BulkVectoriser(model_name, device, model_properties, normalize_embeddings, vectorise_fn=vectorise_fn)
        self._bulk.local_only = True
        # this rank's embeddings: every settled group leaves its [n, D] block(s) WHOLE — on the host in ONE growing slab (the group's D2H lands in
        # pinned memory and is copied once, to the slab's tail), on RCCL ranks as device tensors that never visit the host before the gather — plus,
        # per request, the positions of its rows in the rank's row space (NumPy arithmetic per request, nothing per row)
        self._slab = _RowSlab()
        self._dev_blocks: list = []                           # RCCL ranks: the groups' blocks in HBM, in slab order
        self._nrows = 0                                       # rows filed (slab rows or rows of _dev_blocks)
        self._index: List[Tuple[int, Hashable]] = []          # (request index, key) per filed row, in submission order
        self._perm: List[np.ndarray] = []                     # per filed request: position of each of its keys in the rank's row space
        self._on_device = bool(self._dist) and self.world > 1 and self._dist.get_backend() == "nccl" and str(device).startswith("cuda")
        self._bulk.rows_on_device = self._on_device
        self.touched: List[int] = []                          # request indices this rank was handed (tests: ownership)
        self.failed: List[int] = []                           # owned requests whose encode raised since the last collect()
        self.failed_requests: List[int] = []                  # after collect(): every rank's failed requests (root), own ones elsewhere
        self.errors: Dict[int, BaseException] = {}            # request index -> what its encode raised (since the last collect())
        # merging (0 images = off: every request is its own group)
        self.merge_images = MERGE_IMAGES if merge_images is None else int(merge_images)
        self.merge_text_tokens = MERGE_TEXT_TOKENS if merge_text_tokens is None else int(merge_text_tokens)
        self.merge_max_requests = MERGE_MAX_REQUESTS
        self.merge_deadline_ms = MERGE_DEADLINE_MS if merge_deadline_ms is None else float(merge_deadline_ms)
        self.groups_launched: List[List[int]] = []            # request indices of every group launched since the last collect() (tests, bench)
        self._open: List[Tuple[int, list]] = []               # the group being filled: (request index, items)
        self._open_images = 0
        self._open_tokens = 0.0
        self._open_since = 0.0
        # pipeline_depth groups in flight (default ONE): launching group g tokenises / packs / enqueues it and only THEN copies group
        # g - pipeline_depth's rows to the host, so the host side of a group (Python bookkeeping, tokeniser, Pillow -> pinned staging) overlaps the
        # previous groups' kernels.  pipeline_depth = 0: synchronous.
        self.pipeline_depth = PIPELINE_DEPTH
        self._inflight_q: List[Tuple[List[Tuple[int, list]], PendingFlush]] = []      # launched groups, oldest first
        # submit() / drain() / collect() and the deadline thread all mutate the state above: one re-entrant lock, held across a launch
        self._lock = threading.RLock()
        self._cv = threading.Condition(self._lock)
        self._closed = False
        self._deadline_thread: Optional[threading.Thread] = None
        self.overlap_settle = OVERLAP_SETTLE     # file the leaving group's rows while the next group is being launched (a launcher thread)
        self._launcher = None

    # ---- ownership ------------------------------------------------------------------------------------------------------------------------
    def owner(self, request_index: int) -> int:
        return request_index % self.world

    def owns(self, request_index: int) -> bool:
        return self.owner(request_index) == self.rank

    # ---- bookkeeping ----------------------------------------------------------------------------------------------------------------------
    def _fail(self, request_index: int, error: BaseException) -> None:
        self.failed.append(request_index)
        self.errors[request_index] = error

    @property
    def _rows(self):
        """the filed rows in submission order (tests / diagnostics: the stream itself never touches single rows before collect())"""
        if not self._nrows:
            return []
        parts = [torch_cat_host(self._dev_blocks)] if self._on_device else [c[:u] for c, u in zip(self._slab.chunks, self._slab.used) if u]
        which, local = _rows_at(parts, np.concatenate(self._perm))
        return [parts[b][l] for b, l in zip(which.tolist(), local.tolist())]

    def _place(self, group, base: int, n_text: int) -> None:
        """labels and row positions of a filed group whose rows sit at [base, ...) of the rank's row space as [the group's texts | its images]"""
        t_off, i_off = base, base + n_text
        for _, req in group:
            kinds = np.frombuffer(req.kinds, dtype=np.uint8)
            is_img = kinds.astype(bool)
            pos = np.where(is_img, i_off + np.cumsum(is_img) - 1, t_off + np.cumsum(~is_img) - 1)
            self._perm.append(pos.astype(np.int64, copy=False))
            self._index.extend(req.keys)
            n_img = int(is_img.sum())
            t_off += len(kinds) - n_img
            i_off += n_img

    def _file_blocks(self, group, blocks) -> None:
        """a settled group's blocks ([(modality, items, rows)], texts first) into the rank's row space.  Touching the rows is where an asynchronous
        device fault surfaces: nothing is filed then, the modality is re-queued by the caller's handler (PendingFlush.requeue)"""
        by_mod = {m: (items, rows) for m, items, rows in blocks}
        parts = [by_mod[m] for m in (Modality.TEXT, Modality.IMAGE) if m in by_mod]
        n_text = len(by_mod[Modality.TEXT][0]) if Modality.TEXT in by_mod else 0
        total = sum(len(items) for items, _ in parts)
        if total != sum(len(req.keys) for _, req in group):
            raise RuntimeError("a flush returned a different number of rows than its group queued")
        if self._on_device:
            tensors = [rows.tensor() if isinstance(rows, _DeviceRows) else rows for _, rows in parts]   # (waits for the towers; raises a device fault)
            self._dev_blocks.extend(tensors)
        else:
            D = int(parts[0][1].shape[1])
            tail = self._slab.tail(total, D)
            at = 0
            for items, rows in parts:
                n = len(items)
                if isinstance(rows, np.ndarray):
                    np.copyto(tail[at:at + n], rows)
                else:
                    rows.numpy_into(tail[at:at + n])
                at += n
            self._slab.commit(total)
        self._place(group, self._nrows, n_text)
        self._nrows += total

    def _file(self, group, out) -> None:
        """the same from a {key: row} dictionary (the one-by-one re-run of a failed group: BulkVectoriser.flush())"""
        for request_index, req in group:
            if not req.keys:
                continue
            block = np.stack([out[ck] for ck in req.keys]).astype(np.float32, copy=False)
            n = block.shape[0]
            if self._on_device:
                import torch
                self._dev_blocks.append(torch.from_numpy(block).to(self.device))
            else:
                np.copyto(self._slab.tail(n, block.shape[1]), block)
                self._slab.commit(n)
            self._perm.append(np.arange(self._nrows, self._nrows + n, dtype=np.int64))
            self._index.extend(req.keys)
            self._nrows += n

    def _queue(self, group) -> None:
        """the group's items into the BulkVectoriser's queues; keys are made unique across requests by the request index (done once, in submit)"""
        for _, req in group:
            self._bulk.add_many(req.texts, Modality.TEXT)
            self._bulk.add_many(req.images, Modality.IMAGE)

    def _run_alone(self, group, raise_for: Optional[int] = None) -> Optional[BaseException]:
        """a merged group failed: its requests one at a time, synchronously — the ones that encode are filed (in submission order), the ones
        that raise are recorded.  -> the first error (raised instead when it belongs to request `raise_for`)"""
        first, mine = None, None
        for request_index, items in group:
            try:
                self._queue([(request_index, items)])
                out = self._bulk.flush()
            except BaseException as e:  # noqa: BLE001 - recorded per request; the caller's own one is re-raised below
                self._bulk.reset()
                self._fail(request_index, e)
                first = first or e
                if request_index == raise_for:
                    mine = e
                continue
            self._file([(request_index, items)], out)
        if mine is not None:
            raise mine
        return first

    def _resolve(self, inflight, reraise: bool, deferred: Optional[list] = None) -> None:
        """wait for a launched group's rows and file them; a failure that surfaces here (an asynchronous device error at the deferred copy)
        fails a lone request, and sends a merged group through `_run_alone`.  `deferred` (the overlapped settle: a launcher thread is inside
        flush_async on the SAME BulkVectoriser right now): the failed group is only noted — its recovery resets and re-uses that BulkVectoriser and
        must wait until the launcher has returned (a concurrent reset() could drop the new group's queued items)"""
        group, handle = inflight
        try:
            self._file_blocks(group, handle.result_blocks())
        except BaseException as e:  # noqa: BLE001
            if deferred is not None:
                deferred.append((group, e))
                return
            self._recover(group, e, reraise)

    def _recover(self, group, e: BaseException, reraise: bool) -> None:
        self._bulk.reset()
        if len(group) == 1:
            self._fail(group[0][0], e)
            if reraise:
                raise e
            return
        first = self._run_alone(group)
        if reraise and first is not None:
            raise first

    @property
    def _inflight(self):
        """the youngest group in flight (tests / diagnostics)"""
        return self._inflight_q[-1] if self._inflight_q else None

    def _settle(self, reraise: bool, keep: int = 0, deferred: Optional[list] = None) -> None:
        """file the rows of all but the `keep` youngest groups in flight, oldest first; `reraise`: raise the LAST settled group's error"""
        while len(self._inflight_q) > keep:
            inflight = self._inflight_q.pop(0)
            self._resolve(inflight, reraise and len(self._inflight_q) == keep, deferred)

    def _launch_open(self, raise_for: Optional[int] = None) -> None:
        """(lock held) everything waiting becomes ONE group: queued, tokenised / packed / enqueued now; the previous group's rows are filed
        behind it.  A host-side failure is isolated per request (`_run_alone`)."""
        group, self._open = self._open, []
        self._open_images, self._open_tokens = 0, 0.0
        if not group:
            return
        self.groups_launched.append([r for r, _ in group])
        try:
            self._queue(group)
            if self.overlap_settle and self.pipeline_depth > 0 and self._inflight_q and self._bulk._two_threads():
                # the launch (tokenise / pack / enqueue: milliseconds of host work) runs on a launcher thread while THIS thread files the rows of
                # the group(s) that have to leave the pipeline anyway — the wait for their D2H and the per-row bookkeeping no longer sit in front
                # of the next pack (on a slow host the stream was bound by exactly this thread: profiles/r05t_stream_overlap_settle.txt)
                if self._launcher is None:
                    from concurrent.futures import ThreadPoolExecutor
                    self._launcher = ThreadPoolExecutor(max_workers=1, thread_name_prefix="marqo-amd-ingest-launch")
                fut = self._launcher.submit(self._bulk.flush_async)
                late: list = []
                try:
                    self._settle(reraise=False, keep=self.pipeline_depth - 1, deferred=late)
                finally:
                    try:
                        handle = fut.result()
                    finally:
                        # groups whose rows failed at the deferred copy: re-run request by request NOW — the launcher is out of the
                        # BulkVectoriser (its group's items sit in `handle`, not in the queue `_recover` resets), the order of filing is kept
                        # (the new group is filed later)
                        for g_, e_ in late:
                            self._recover(g_, e_, reraise=False)
            else:
                handle = self._bulk.flush_async()
        except BaseException as e:
            # One bad document (an undecodable image) must not poison this rank's stream: BulkVectoriser re-queues the failed modality and
            # keeps the other one's rows for a retry, but a request is all-or-nothing here — drop both, so that the NEXT group starts from
            # an empty queue.  (The group in flight is not touched: its rows are in its handle, not in the queue that is reset here.)
            self._bulk.reset()
            if len(group) == 1:
                self._fail(group[0][0], e)
                if raise_for == group[0][0]:
                    raise
                return
            self._settle(reraise=False)            # rows are filed in submission order: the group in flight first
            self._run_alone(group, raise_for)
            return
        self._inflight_q.append((group, handle))
        if self.pipeline_depth <= 0:
            self._settle(reraise=raise_for is not None and len(group) == 1)
        else:
            self._settle(reraise=False, keep=self.pipeline_depth)

    def _open_is_full(self) -> bool:
        return (self.merge_images <= 0 or self._open_images >= self.merge_images or self._open_tokens >= self.merge_text_tokens
                or len(self._open) >= self.merge_max_requests)

    # ---- the deadline: a waiting request is never held longer than merge_deadline_ms by a slow producer ----------------------------------------
    def _deadline_tick(self) -> None:
        with self._cv:
            if not self._open:
                self._cv.wait(0.05)
                return
            left = self._open_since + self.merge_deadline_ms * 1e-3 - _now()
            if left > 0:
                self._cv.wait(left)
                return
            try:
                self._launch_open()
            except BaseException:  # noqa: BLE001 - recorded per request by _launch_open; nobody to raise to on this thread
                pass

    def _ensure_deadline_thread(self) -> None:
        if self._deadline_thread is None and self.merge_images > 0 and self.merge_deadline_ms > 0:
            import weakref
            self._deadline_thread = threading.Thread(target=_deadline_loop, args=(weakref.ref(self),), name="marqo-amd-ingest-deadline", daemon=True)
            self._deadline_thread.start()

    def close(self) -> None:
        """stop the deadline thread (an object that is simply dropped stops it too, within 50 ms of being collected)"""
        with self._cv:
            self._closed = True
            self._cv.notify_all()

    # ---- the stream -----------------------------------------------------------------------------------------------------------------------
    def drain(self) -> None:
        """launch what is waiting and wait for everything launched; raises what the LAST group's encode raised (a merged group: its first
        failing request's error) — every failure is recorded in `failed` / `errors` either way"""
        with self._lock:
            pending_error = None
            try:
                self._launch_open()
            except BaseException as e:  # noqa: BLE001
                pending_error = e
            self._settle(reraise=True)
            if pending_error is not None:
                raise pending_error

    def submit(self, request_index: int, items) -> None:
        """hand over one owned request (its rows stay on this rank until collect()).  The request joins the group being filled; when that
        makes the group chip-filling (or merging is off) the group is launched NOW: everything that can fail on the HOST — decoding,
        tokenising, staging, the enqueue itself — fails here; a failure of request i itself is raised, other requests' failures are only
        recorded.  Device work stays in flight until the next launch / drain() / collect(): an asynchronous device error is recorded
        (`failed`, `errors[i]`, collect()'s `failed_requests`) then, without failing that call; drain() after submit() is the synchronous form."""
        if not self.owns(request_index):
            raise ValueError(f"rank {self.rank} was handed request {request_index}, which belongs to rank {self.owner(request_index)}")
        # ONE pass over the request's items (this thread is the stream's critical path: it also packs the images): per-modality (key, content) lists
        # as the BulkVectoriser queues them, the (request, key) labels in submission order, the merge counters
        req = _Request()
        n_tokens = 0.0
        for key, content, m in items:
            ck = (request_index, key)
            req.keys.append(ck)
            if m == Modality.TEXT:
                req.texts.append((ck, content))
                req.kinds.append(0)
                n_tokens += 2.0 + len(content) / 4.0 if isinstance(content, str) else 1.0     # (= estimate_tokens)
            elif m == Modality.IMAGE:
                req.images.append((ck, content))
                req.kinds.append(1)
            else:
                raise ValueError(f"unsupported modality {m}")
        items = req
        n_images = len(req.images)
        with self._cv:
            self.touched.append(request_index)
            if not self._open:
                self._open_since = _now()
            self._open.append((request_index, items))
            self._open_images += n_images
            self._open_tokens += n_tokens
            if self._open_is_full():
                self._launch_open(raise_for=request_index)
            else:
                self._ensure_deadline_thread()
                self._cv.notify_all()

    def collect(self) -> Dict[int, Dict[Hashable, np.ndarray]]:
        """gather every rank's rows on the root -> {request index: {key: row}} there, {} elsewhere; resets the store.
        What travels: ONE [n_r, D] tensor per rank — the rank's slab as it stands, or, on RCCL ranks, the concatenation IN HBM of the groups' device
        blocks (the rows' first host is the root, behind the gather) — plus, as Python objects, the (request, key) labels and one int64 position per
        label.  A non-root rank runs no per-row Python here or anywhere before: its work in collect() is the all_gather_object of its labels and the
        gather."""
        import torch
        with self._lock:
            try:
                self._launch_open()
            except BaseException:  # noqa: BLE001 - reported through failed_requests; every rank must reach the collective
                pass
            self._settle(reraise=False)
            index, perms, n = self._index, self._perm, self._nrows
            local = self._slab.take()
            dev_blocks, self._dev_blocks = self._dev_blocks, []
            self._index, self._perm, self._nrows = [], [], 0
            failed, self.failed = self.failed, []
            self.groups_launched = []
            self.errors = {i: e for i, e in self.errors.items() if i in failed}   # kept until the NEXT collect() for the caller to inspect
        self.failed_requests = sorted(failed)
        perm = np.concatenate(perms) if perms else np.zeros(0, dtype=np.int64)

        def scatter(out, labels, positions, parts):
            which, at = _rows_at(parts, positions)
            cur_ri, cur = None, None
            for (ri, key), b, l in zip(labels, which.tolist(), at.tolist()):
                if ri != cur_ri:
                    cur_ri, cur = ri, out.setdefault(ri, {})
                cur[key] = parts[b][l]
            return out
        if self.world == 1:
            # (no copy: the rows are views of the slab's chunks)
            return scatter({}, index, perm, local) if n else {}
        dist = self._dist
        where = self.device if self._on_device else "cpu"
        # who holds what: (row count, width) per rank + the labels and their row positions (tiny next to the rows)
        if self._on_device:
            t = torch.cat(dev_blocks) if len(dev_blocks) > 1 else (dev_blocks[0] if dev_blocks else None)
        else:
            t = torch.from_numpy(local[0] if len(local) == 1 else np.concatenate(local)) if n else None      # (the ONE host copy a multi-chunk rank pays, at the end of the stream)
        meta = [None] * self.world
        mine = (n, int(t.shape[1]) if t is not None else 0, (index, perm) if self.rank != self.root else None, failed)
        dist.all_gather_object(meta, mine)
        if self.rank == self.root:
            self.failed_requests = sorted(i for m in meta for i in m[3])
        D = max(m[1] for m in meta)
        counts = [m[0] for m in meta]
        if sum(counts) == 0:
            return {}
        if t is None:
            t = torch.zeros(0, D, dtype=torch.float32, device=where)
        full = gather_embeddings_to_root(t.to(dtype=torch.float32), counts, root=self.root)
        if self.rank != self.root:
            return {}
        full = full.cpu().numpy()          # the ONE device -> host copy of the stream's rows (RCCL ranks)
        out: Dict[int, Dict[Hashable, np.ndarray]] = {}
        base = 0
        for r in range(self.world):
            labels, positions = (index, perm) if r == self.root else meta[r][2]
            scatter(out, labels, positions + base, [full])
            base += counts[r]
        return out


def torch_cat_host(blocks) -> np.ndarray:
    import torch
    return torch.cat(blocks).cpu().numpy() if blocks else np.zeros((0, 0), dtype=np.float32)
