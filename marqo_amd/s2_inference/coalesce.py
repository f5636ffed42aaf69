"""Transparent cross-request micro-batching inside `vectorise()` (on by default for small calls since round 4, see `window_for`;
MARQO_AMD_COALESCE_US=0 = the reference's behaviour).

Why: the reference's default add_documents handlers call `vectorise` once per document per field — N = 1..10 items per call
(src/marqo/core/vespa_index/add_documents_handler.py:264-290,307-342, src/marqo/core/inference/tensor_fields_container.py:179-223) — from up
to 8 indexing + 8 search request threads at once (src/marqo/api/configs.py:27-28).  `BulkVectoriser` only helps callers that are rewritten
to use it; an UNMODIFIED Marqo on PER_DOCUMENT would run a 0.4-0.6 ms small-call tower pass per call, serially per thread, and the GPU
would see a few dozen rows at a time.

What: concurrent calls for the same (model cache key, modality, normalize, keyword arguments) are merged into ONE engine `encode` call and
de-multiplexed afterwards.  The batching is "natural": a call that finds the engine idle for its key runs at once (no added latency for a
lone caller); calls that arrive while one is executing queue up in an open group whose first member (the leader) fires as soon as the
engine is free again — or after MARQO_AMD_COALESCE_US at the latest — with everything that joined meanwhile.  The embeddings are those of
the merged call: rows of a batch are independent on this engine (per-row normalisation, no cross-item reduction), bit-identical to the
un-merged call as long as both take the same GEMM kernel family (tests/test_coalesce.py bounds the difference for the small-M family).

Errors: when a merged call raises, no participant can tell whose item was at fault, so every participant re-runs its OWN request alone:
the faulty request raises its own exception in its own thread, the others get their embeddings.
"""
from __future__ import annotations

import os
import threading
import time
from typing import Any, Callable, Dict, Hashable, List, Optional


DEFAULT_WINDOW_US = 1000.0
SMALL_CALL_ITEMS = 16


def window_seconds() -> float:
    """MARQO_AMD_COALESCE_US: the longest a leader waits for the engine to come free / for others to join, in microseconds; 0 = off;
    unset = DEFAULT_WINDOW_US (for the small calls `window_for` lets in)"""
    v = os.environ.get("MARQO_AMD_COALESCE_US", "")
    try:
        return max(0.0, float(v)) * 1e-6 if v else DEFAULT_WINDOW_US * 1e-6
    except ValueError:
        return 0.0


def window_for(n_items: int) -> float:
    """the window a call of n_items gets.  Default (MARQO_AMD_COALESCE_US unset): calls of <= SMALL_CALL_ITEMS items (the per-document,
    per-field calls of an unmodified Marqo, MARQO_AMD_COALESCE_SMALL_ITEMS to change) coalesce, larger ones never wait for anybody.  A lone
    caller is not delayed either way (natural batching: a call that finds fewer than `depth` calls of its key executing fires at once).
    MARQO_AMD_COALESCE_US=<us> set explicitly: every call of <= MARQO_AMD_COALESCE_MAX_ITEMS items takes part; =0: off (the reference's behaviour)."""
    if os.environ.get("MARQO_AMD_COALESCE_US", ""):
        return window_seconds() if n_items <= max_items() else 0.0
    try:
        small = max(0, int(os.environ.get("MARQO_AMD_COALESCE_SMALL_ITEMS", str(SMALL_CALL_ITEMS))))
    except ValueError:
        small = SMALL_CALL_ITEMS
    return DEFAULT_WINDOW_US * 1e-6 if n_items <= small else 0.0


def explicit() -> bool:
    """MARQO_AMD_COALESCE_US set by the operator (to anything): the default's exclusions below no longer apply"""
    return bool(os.environ.get("MARQO_AMD_COALESCE_US", ""))


def fetches_content(batch, is_text: bool) -> bool:
    """non-text items given as strings are URLs / paths (Marqo's search path hands image-URL queries straight to vectorise): the engine call
    DOWNLOADS them.  A merged call would do every participant's downloads one after the other on the leader's thread while the followers
    block — before, every request thread downloaded in parallel — and one slow or failing URL would make the whole group re-run alone and
    download twice.  Such calls stay out of the default coalescing (they still merge when MARQO_AMD_COALESCE_US is set explicitly)."""
    return (not is_text) and any(isinstance(item, (str, bytes)) for item in batch)


def max_items() -> int:
    """MARQO_AMD_COALESCE_MAX_ITEMS: a merged call carries at most this many items (larger requests are chip-filling on their own)"""
    try:
        return max(1, int(os.environ.get("MARQO_AMD_COALESCE_MAX_ITEMS", "512")))
    except ValueError:
        return 512


def depth() -> int:
    """MARQO_AMD_COALESCE_DEPTH: merged engine calls of one key that may be in flight at once.  2 (default): the next group's leader fires
    while ONE call is still executing, so its host part (tokenising / packing / enqueue) overlaps the GPU part of the call before it"""
    try:
        return max(1, int(os.environ.get("MARQO_AMD_COALESCE_DEPTH", "2")))
    except ValueError:
        return 2


class _Group:
    __slots__ = ("parts", "n", "open", "done", "results", "error", "ready")

    def __init__(self):
        self.parts: List[list] = []
        self.n = 0
        self.open = True
        self.done = threading.Event()
        self.results: Optional[list] = None
        self.error: Optional[BaseException] = None
        self.ready = None          # device results: event recorded on the leader's stream behind the merged call


def _device_event(out):
    """merged rows that stay in HBM (`vectorise_device`) are written on the LEADER's stream: an event behind them, for the followers"""
    if hasattr(out, "is_cuda") and out.is_cuda:
        import torch
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(out.device))
        return ev
    return None


def _own_rows(rows, ready):
    """a participant's copy of its slice.  Device rows: the copy runs on the participant's current stream, which first waits for the leader's
    event; the shared block is marked as used by that stream so that the allocator does not hand it out again before the copy has run"""
    if hasattr(rows, "clone"):
        if ready is not None and rows.is_cuda:
            import torch
            cur = torch.cuda.current_stream(rows.device)
            cur.wait_event(ready)
            rows.record_stream(cur)
        return rows.clone()
    return rows.copy()


class Coalescer:
    def __init__(self):
        self._lock = threading.Condition()
        self._groups: Dict[Hashable, _Group] = {}
        self._busy: Dict[Hashable, int] = {}
        self.stats = {"calls": 0, "merged_calls": 0, "engine_calls": 0}   # (tests / diagnostics)

    def submit(self, key: Hashable, content: list, run: Callable[[list], Any], window: float, limit: int):
        """run(content) -> [len(content), D] array (ndarray or device tensor); returns this caller's rows"""
        n = len(content)
        with self._lock:
            self.stats["calls"] += 1
            g = self._groups.get(key)
            if g is not None and g.open and g.n + n <= limit:
                slot, leader = len(g.parts), False
            else:
                g = _Group()
                self._groups[key] = g
                slot, leader = 0, True
            g.parts.append(content)
            g.n += n
            if not leader and g.n >= limit:
                self._lock.notify_all()          # the group is full: the leader need not wait any longer
        if not leader:
            g.done.wait()
        else:
            deadline = time.perf_counter() + window
            with self._lock:
                # natural batching: fire as soon as fewer than `depth` calls of this key are executing; until then wait (others join), `window` at most
                d = depth()
                while self._busy.get(key, 0) >= d and g.n < limit:
                    left = deadline - time.perf_counter()
                    if left <= 0:
                        break
                    self._lock.wait(left)
                g.open = False
                if self._groups.get(key) is g:
                    del self._groups[key]
                self._busy[key] = self._busy.get(key, 0) + 1
                self.stats["engine_calls"] += 1
                if len(g.parts) > 1:
                    self.stats["merged_calls"] += len(g.parts)
            try:
                merged = g.parts[0] if len(g.parts) == 1 else [item for part in g.parts for item in part]
                out = run(merged)
                if len(out) != g.n:
                    raise RuntimeError(f"vectorise returned {len(out)} embeddings for {g.n} items")
                pos, res = 0, []
                for part in g.parts:
                    res.append(out[pos:pos + len(part)])
                    pos += len(part)
                g.ready = _device_event(out) if len(g.parts) > 1 else None
                g.results = res
            except BaseException as e:  # noqa: BLE001 - handed to every participant below
                g.error = e
            finally:
                with self._lock:
                    self._busy[key] -= 1
                    if self._busy[key] <= 0:
                        del self._busy[key]
                    self._lock.notify_all()
                g.done.set()
        if g.error is not None:
            if len(g.parts) == 1:
                raise g.error
            return run(content)       # whose item was it?  everyone re-runs alone: only the faulty request raises
        rows = g.results[slot]
        if len(g.parts) > 1:
            rows = _own_rows(rows, g.ready)   # callers own their result (no views into a shared matrix)
        return rows


_coalescer = Coalescer()


def get_coalescer() -> Coalescer:
    return _coalescer
