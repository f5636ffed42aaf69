"""Python face of the native request queue (`mq_queue_*`, csrc/queue.hip, ABI 14): concurrent small text calls of one tower are merged into ONE tower
call on worker threads that never hold the interpreter lock.

Where the reference has this load: up to 8 indexing + 8 search request threads (reference: src/marqo/api/configs.py:27-28), each calling
`vectorise()` with one query or the chunks of one document field (src/marqo/core/inference/tensor_fields_container.py:179-223).  A caller tokenises
its own texts on its own thread and blocks inside `mq_queue_encode` (ctypes drops the GIL around the call); packing, H2D, the ~100 launches of the tower
pass, D2H and the wake-up are native.  The Python-level coalescer (s2_inference/coalesce.py) stays for everything that has no queue (image calls, loaders
without an engine tower) and steps aside for text calls of a tower that has one.

MARQO_AMD_NATIVE_QUEUE=0 turns the queue off (the coalescer then merges text calls as before round 6); _SEQS / _DEPTH / _WINDOW_US size it, _GRAPHS=0 keeps
lone single queries on the tower's own captured graph."""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Dict, Optional

import numpy as np

from marqo_amd import _lib as L


def _env_int(name: str, default: int, lo: int, hi: int) -> int:
    try:
        return min(hi, max(lo, int(os.environ.get(name, str(default)))))
    except ValueError:
        return default


ENABLED = os.environ.get("MARQO_AMD_NATIVE_QUEUE", "1") != "0"
# a request of more sequences than this goes the direct way (it fills enough of the chip on its own, and the merged call's staging is sized by it)
MAX_SEQS = _env_int("MARQO_AMD_NATIVE_QUEUE_SEQS", 64, 1, 4096)
# token rows per merged call (and per request): bounds the queue's device scratch (sized once, for the largest call) on towers with long contexts —
# 64 sequences x 512 positions would reserve a 32 k-row workspace per (tower, normalize) for calls that are a few hundred rows in practice
MAX_ROWS = _env_int("MARQO_AMD_NATIVE_QUEUE_ROWS", 16384, 64, 1 << 18)
DEPTH = _env_int("MARQO_AMD_NATIVE_QUEUE_DEPTH", 2, 1, 4)
# with DEPTH > 1: the lanes beyond the first are HELPERS — they take a backlog of at most this many sequences, and only while the group the first lane is
# running is that small too (light load runs side by side: 2 request threads 1.00 -> 0.61 ms per one-query call; heavy load keeps one lane's large groups:
# unchanged within noise; profiles/r08t_queue_helper_lanes_sweep.txt); 0 = every lane takes whatever waits (measured worse under heavy load, r08d)
HELPER_SEQS = _env_int("MARQO_AMD_NATIVE_QUEUE_HELPER_SEQS", 4, 0, 64)
WINDOW_US = _env_int("MARQO_AMD_NATIVE_QUEUE_WINDOW_US", 0, 0, 100000)
# a LONE single query goes through the queue too and replays a hipGraph captured on the worker (0 = it keeps the tower's own captured graph, through torch)
GRAPHS = os.environ.get("MARQO_AMD_NATIVE_QUEUE_GRAPHS", "1") != "0"


# image towers: a request of at most this many preprocessed images goes through the queue (the per-document, per-field calls of an unmodified Marqo:
# 1-4 tensors from `.preprocess`); a merged call carries up to IMAGE_MAX_SEQS.  Larger lists are the slab path's (open_clip_model._encode_image).
IMAGE_REQUEST_MAX = _env_int("MARQO_AMD_NATIVE_QUEUE_IMAGE_REQUEST", 8, 0, 64)
IMAGE_MAX_SEQS = _env_int("MARQO_AMD_NATIVE_QUEUE_IMAGE_SEQS", 32, 1, 256)


def gone(e: Exception) -> bool:
    """a queue call that failed because the queue was closed under it (the tower re-creates its queue when its policy fields change): the caller takes
    the regular path for this one request"""
    msg = str(e)
    return "null queue" in msg or "being destroyed" in msg


class TextQueue:
    """one `mq_queue` of a text tower for one value of `normalize`.  `cfg` / `w` are the tower's ctypes structs (kept alive here: the queue reads them
    at every call); thread-safe; recreated in a fork()ed child (worker threads do not survive a fork)."""

    def __init__(self, lib, kind: int, cfg, w, device_index: int, out_dim: int, max_len: int, normalize: bool,
                 max_seqs: int = 0, depth: int = 0, window_us: Optional[int] = None, graphs: Optional[bool] = None):
        self._lib, self._kind, self._cfg, self._w = lib, kind, cfg, w
        self.out_dim, self.max_len = int(out_dim), int(max_len)
        self.max_seqs = int(max_seqs or MAX_SEQS)
        self.max_rows = max(self.max_len, min(self.max_seqs * self.max_len, MAX_ROWS))
        self._qcfg = L.QueueCfg(kind=kind, device=int(device_index), max_seqs=self.max_seqs, max_rows=self.max_rows, normalize=1 if normalize else 0,
                                depth=int(depth or DEPTH), window_us=int(WINDOW_US if window_us is None else window_us),
                                graphs=1 if (GRAPHS if graphs is None else graphs) else 0, helper_seqs=min(HELPER_SEQS, self.max_seqs), reserved=0)
        self._h = C.c_void_p()
        self._pid = os.getpid()
        self._lock = threading.Condition()
        self._inflight, self._closing = 0, False
        L.check(lib.mq_queue_create(C.byref(self._qcfg), C.cast(C.byref(cfg), C.c_void_p), C.cast(C.byref(w), C.c_void_p), C.byref(self._h)), "mq_queue_create")

    def takes(self, nseq: int, rows: int) -> bool:
        return 1 <= nseq <= self.max_seqs and rows <= self.max_rows

    def _handle(self) -> C.c_void_p:
        if self._pid != os.getpid():        # fork()ed child: the parent's workers are not here; a queue of our own (the parent's handle is left alone)
            with self._lock:
                if self._pid != os.getpid():
                    h = C.c_void_p()
                    L.check(self._lib.mq_queue_create(C.byref(self._qcfg), C.cast(C.byref(self._cfg), C.c_void_p), C.cast(C.byref(self._w), C.c_void_p),
                                                      C.byref(h)), "mq_queue_create")
                    self._h, self._pid, self._inflight, self._closing = h, os.getpid(), 0, False
        return self._h

    def _call(self, fn, *args) -> None:
        """one foreign call on the live queue.  `close()` waits for the calls that are inside (they finish by themselves: the workers keep serving until
        the queue is destroyed) and refuses new ones, so the handle is never used after mq_queue_destroy has started"""
        h = self._handle()
        with self._lock:
            if self._closing or not h:
                raise L.MarqoHipError("mq_queue: null queue (closed)")
            self._inflight += 1
        try:
            L.check(fn(h, *args), fn.__name__)
        finally:
            with self._lock:
                self._inflight -= 1
                if self._inflight == 0:
                    self._lock.notify_all()

    def encode(self, packed_ids: np.ndarray, lengths: np.ndarray) -> np.ndarray:
        """packed_ids int32 [sum(lengths)], lengths int32 [n] -> fp32 [n, out_dim] (host).  Blocks (GIL released) until the merged call that carried
        this request has finished."""
        ids = np.ascontiguousarray(packed_ids, dtype=np.int32)
        lens = np.ascontiguousarray(lengths, dtype=np.int32)
        n = int(lens.size)
        if int(lens.sum()) != int(ids.size):
            raise ValueError(f"packed ids hold {ids.size} tokens, the lengths add up to {int(lens.sum())}")
        out = np.empty((n, self.out_dim), dtype=np.float32)
        if n:
            self._call(self._lib.mq_queue_encode, ids.ctypes.data, lens.ctypes.data, n, out.ctypes.data)
        return out

    def encode_raw(self, ids32: np.ndarray, lens32: np.ndarray, n: int) -> np.ndarray:
        """`encode` for callers that vouch for their arrays (contiguous int32, lengths adding up to the ids): the towers' per-request path, where
        every NumPy call is interpreter time that 16 request threads queue up for"""
        out = np.empty((n, self.out_dim), dtype=np.float32)
        self._call(self._lib.mq_queue_encode, ids32.ctypes.data, lens32.ctypes.data, n, out.ctypes.data)
        return out

    def stats(self) -> Dict[str, int]:
        st = L.QueueStats()
        self._call(self._lib.mq_queue_get_stats, C.byref(st))
        return {name: int(getattr(st, name)) for name, _ in L.QueueStats._fields_}

    def close(self) -> None:
        with self._lock:
            if self._closing:
                return
            self._closing = True
            while self._inflight > 0:
                self._lock.wait(1.0)
            h, self._h = self._h, C.c_void_p()
        if h and self._pid == os.getpid():
            self._lib.mq_queue_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ImageQueue(TextQueue):
    """one `mq_queue` (MQ_QUEUE_IMAGE_F32) of an image tower for one value of `normalize`: a request = device pointers of preprocessed fp32 [3, S, S]
    images (the views `.preprocess` hands out), gathered with the other callers' into one `mq_encode_image_f32` per group"""

    def __init__(self, lib, cfg, w, device_index: int, out_dim: int, normalize: bool, max_seqs: int = 0, depth: int = 0, graphs: Optional[bool] = None):
        self._lib, self._kind, self._cfg, self._w = lib, L.QUEUE_IMAGE_F32, cfg, w
        self.out_dim, self.max_len = int(out_dim), 1
        self.max_seqs = int(max_seqs or IMAGE_MAX_SEQS)
        self.max_rows = self.max_seqs
        self._qcfg = L.QueueCfg(kind=L.QUEUE_IMAGE_F32, device=int(device_index), max_seqs=self.max_seqs, max_rows=self.max_rows, normalize=1 if normalize else 0,
                                depth=int(depth or DEPTH), window_us=int(WINDOW_US), graphs=1 if (GRAPHS if graphs is None else graphs) else 0, helper_seqs=min(HELPER_SEQS, self.max_seqs), reserved=0)
        self._h = C.c_void_p()
        self._pid = os.getpid()
        self._lock = threading.Condition()
        self._inflight, self._closing = 0, False
        L.check(lib.mq_queue_create(C.byref(self._qcfg), C.cast(C.byref(cfg), C.c_void_p), C.cast(C.byref(w), C.c_void_p), C.byref(self._h)), "mq_queue_create")

    def encode_ptrs(self, ptrs) -> np.ndarray:
        """ptrs: device addresses of n complete fp32 [3, S, S] images -> fp32 [n, out_dim] (host); blocks with the GIL released"""
        n = len(ptrs)
        out = np.empty((n, self.out_dim), dtype=np.float32)
        if n:
            arr = (C.c_void_p * n)(*ptrs)
            self._call(self._lib.mq_queue_encode_images, arr, n, out.ctypes.data)
        return out

    def encode(self, *a, **k):
        raise TypeError("an image queue takes device pointers: encode_ptrs")

    encode_raw = encode
