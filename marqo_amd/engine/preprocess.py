"""Device-side image preprocessing (K10 / K11): host packing + the C-ABI calls of csrc/preprocess.hip.

Replaces the per-image PIL work the reference does in Python download threads
(src/marqo/s2_inference/clip_utils.py:48-67 through add_docs.py:121-134, and
src/marqo/s2_inference/processing/image.py:120-151): raw decoded uint8 pixels of a whole batch go to the
GPU in ONE pinned H2D copy and are resized / cropped / chunked there, bit-identically to Pillow.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from marqo_amd import _lib as L
from marqo_amd.engine.archs import OPENAI_DATASET_MEAN, OPENAI_DATASET_STD

try:  # Pillow >= 11.3 exports its image memory through the Arrow C interface; pyarrow is the consumer
    import pyarrow as _arrow
except Exception:  # pragma: no cover
    _arrow = None

ArrayLike = Union[np.ndarray, torch.Tensor, "Rgbx", "Rgba", "NearestRgb"]


# pieces of a pack -> H2D pipeline (1, the default: pack everything, then one copy).  Measured with 4 pieces (profiles/r02ab_pack_chunks_ab.txt):
# uint8 arrays -1.5 % with one caller and -18 % with four (four times the native calls and copies per request), Pillow images +7 % with
# one caller, neutral with four -> off
PACK_CHUNKS = max(1, int(os.environ.get("MARQO_AMD_PACK_CHUNKS", "1")))
# Pillow batches (the native stager, csrc/py_stage.cpp): images per slice.  A batch of >= 2 slices is staged slice by slice, the pinned H2D copy of
# slice c running on the copy engine while the memcpy threads (GIL released) pack slice c + 1 — the single synchronous caller's 0.9 ms of H2D
# no longer follows its 2.2 ms of packing (profiles/r03i_e2e_phases_1thread.txt): 6.57 -> 5.55 ms per 256-image call, 39.0 k -> 46.2 k
# embeddings/s (profiles/r04d_e2e_slices.txt; 32-image slices 6.22 ms); the tower still runs once on the whole batch.  0 = one piece.
PACK_SLICE = max(0, int(os.environ.get("MARQO_AMD_PACK_SLICE", "64")))
# (slicing one level up — pack -> H2D -> unpack -> resize per slice — measured no better: 5.49 ms with two 128-image slices against 5.55 ms for
# the 64-image pack slices alone, 5.97 ms with four, profiles/r04e_e2e_slices.txt: the tail of the call is the tower, not the resize)
PACK_THREADS = int(os.environ.get("MARQO_AMD_PACK_THREADS", str(max(1, min(8, (os.cpu_count() or 2) // 2)))))   # memcpy threads of a pack


def _export_rgbx(img):
    """Pillow RGB image -> uint8 [H, W, 4] zero-copy view through its Arrow export, or None (no pyarrow, an image stored in several
    blocks, any export quirk)"""
    if _arrow is None or not hasattr(img, "__arrow_c_array__") or img.width <= 0 or img.height <= 0:
        return None
    try:
        flat = _arrow.array(img).flatten().to_numpy(zero_copy_only=True)
        if flat.dtype == np.uint8 and flat.size == img.height * img.width * 4:
            return flat.reshape(img.height, img.width, 4)
    except Exception:  # the plain path is always right
        pass
    return None


class Rgbx:
    """A decoded Pillow RGB image as it sits in memory: 4 bytes per pixel (R, G, B, pad).  `shape` is the logical (H, W, 3).  The bytes are
    staged as they are and repacked on the device (mq_unpack_rgbx): Image.tobytes / np.asarray spend 80-300 us per 224 x 224 image on the
    4 -> 3 byte repack.  `image` is the PIL image; PackedImages hands whole batches of them to the native stager (_mq_stage.gather_rgbx: one
    call, copies with the GIL released).  `.view` is the uint8 [H, W, 4] array for everybody else, exported on first use (~10 us through
    pyarrow; an RGBX copy when Pillow cannot export the image zero-copy)."""
    __slots__ = ("image", "shape", "_view")

    def __init__(self, image, view: Optional[np.ndarray] = None) -> None:
        self.image, self._view = image, view
        self.shape = (image.height, image.width, 3)

    @property
    def view(self) -> np.ndarray:
        if self._view is None:
            v = _export_rgbx(self.image)
            self._view = v if v is not None else np.asarray(self.image.convert("RGBX"))
        return self._view


class Rgba:
    """Pixels of an RGBA / LA image with real transparency: uint8 [H, W, 4].  The reference's transform resizes such an image in its OWN
    mode — Pillow premultiplies by alpha, resamples all four bands and un-premultiplies — and only then drops the alpha band
    (`.convert("RGB")`, clip_utils.py:61-64): flattening first gives different colours wherever alpha < 255.  `shape` is the logical (H, W, 3)."""
    __slots__ = ("array", "shape")

    def __init__(self, array: np.ndarray) -> None:
        self.array = np.ascontiguousarray(array)
        self.shape = (array.shape[0], array.shape[1], 3)


class NearestRgb:
    """RGB pixels (uint8 [H, W, 3]) of a palette ("P") or bilevel ("1") image: Pillow resizes those modes with NEAREST whatever filter the
    transform asks for (Image.resize), and nearest sampling commutes with the palette lookup, so the lookup is done first."""
    __slots__ = ("array", "shape")

    def __init__(self, array: np.ndarray) -> None:
        self.array = np.ascontiguousarray(array)
        self.shape = tuple(array.shape)


def flatten_pixels(px):
    """any pixel container -> something the 3-channel RGB paths take (chunk grids; the alpha band is dropped as `.convert("RGB")` does)"""
    if isinstance(px, Rgba):
        return np.ascontiguousarray(px.array[..., :3])
    if isinstance(px, NearestRgb):
        return px.array
    return px


def pil_pixels(img):
    """PIL image -> the pixel container the GPU preprocessing packs, by MODE, so that the device resize follows what Pillow does for
    that mode: RGB -> Rgbx view (when this Pillow / pyarrow pair exports one) or uint8 [H, W, 3]; RGBA / LA with transparency -> Rgba;
    P / 1 -> NearestRgb; every other mode (L, CMYK, I, F, ...) -> uint8 [H, W, 3] via convert("RGB") — exact for L (the filter is per
    band), flatten-first for the rest."""
    mode = img.mode
    if mode in ("RGBA", "LA"):
        a = np.asarray(img if mode == "RGBA" else img.convert("RGBA"))
        if a.size and int(a[..., 3].min()) == 255:   # opaque: the premultiplied round trip is the identity, take the RGB path
            return np.ascontiguousarray(a[..., :3])
        return Rgba(a)
    if mode in ("P", "1"):
        return NearestRgb(np.asarray(img.convert("RGB")))
    if mode == "RGB" and img.width > 0 and img.height > 0 and hasattr(img, "__arrow_c_array__"):
        if L.load_stage() is not None:     # exported in bulk by the native stager when the batch is packed
            return Rgbx(img)
        v = _export_rgbx(img)
        if v is not None:
            return Rgbx(img, v)
    return np.asarray(img if img.mode == "RGB" else img.convert("RGB"))


def pil_rgb_sizes(images):
    """(heights, widths) int32 arrays when `images` is a non-empty list of loaded Pillow RGB images and the native stager is there — the batch
    fast path of PackedImages (one native scan instead of a Python loop) — else None"""
    stager = L.load_stage()
    if stager is None or not isinstance(images, list) or not images or not hasattr(stager, "rgb_sizes"):
        return None
    h, w = np.empty(len(images), dtype=np.int32), np.empty(len(images), dtype=np.int32)
    return (h, w) if stager.rgb_sizes(images, h, w) else None


def _as_u8_hwc(img, channels: int = 3):
    if isinstance(img, Rgbx):
        return img
    if isinstance(img, np.ndarray):
        if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != channels:
            raise ValueError(f"expected a uint8 [H, W, {channels}] image, got {img.dtype} {tuple(img.shape)}")
        return img
    if img.dtype != torch.uint8 or img.ndim != 3 or img.shape[2] != channels:
        raise ValueError(f"expected a uint8 [H, W, {channels}] image, got {img.dtype} {tuple(img.shape)}")
    return img.contiguous()


def _kind(img) -> int:
    """MQ_IMG_* mode of a pixel container (include/marqo_hip.h)"""
    return L.MQ_IMG_RGBA if isinstance(img, Rgba) else L.MQ_IMG_NEAREST if isinstance(img, NearestRgb) else L.MQ_IMG_RGB


def _align256(v):
    return (v + 255) // 256 * 256


# Host -> device copies of a pack run on a copy stream of the calling thread, not on its compute stream: the copy engine then moves request
# i + 1's pixels while request i's kernels are still running (a pipelined caller — RequestShardedIngest keeps one request in flight — otherwise
# queues the 19 MB / 0.4 ms transfer BEHIND the previous request's towers).  The destination is allocated on the copy stream (the caching
# allocator orders a block's reuse per stream) and handed to the compute stream with an event + record_stream.  MARQO_AMD_COPY_STREAM=0: off.
COPY_STREAM = os.environ.get("MARQO_AMD_COPY_STREAM", "1") != "0"
_copy_tls = threading.local()


class _H2D:
    """`with _H2D(device) as h:` — inside, torch allocations and copies go to the thread's copy stream; on exit the compute stream (the
    stream that was current outside) waits for them.  `h.hand_over(t)` marks a tensor allocated inside as used by the compute stream."""

    def __init__(self, device: torch.device):
        self.on = COPY_STREAM and torch.cuda.is_available() and torch.device(device).type == "cuda"
        self.device = torch.device(device)

    def __enter__(self):
        if not self.on:
            return self
        streams = getattr(_copy_tls, "streams", None)
        if streams is None:
            streams = _copy_tls.streams = {}
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        cs = streams.get(idx)
        if cs is None:
            cs = streams[idx] = torch.cuda.Stream(torch.device("cuda", idx))
        self.compute = torch.cuda.current_stream(torch.device("cuda", idx))
        self.copy = cs
        self._ctx = torch.cuda.stream(cs)
        self._ctx.__enter__()
        return self

    def hand_over(self, t: torch.Tensor) -> torch.Tensor:
        if self.on:
            t.record_stream(self.compute)
        return t

    def __exit__(self, *exc):
        if self.on:
            self._ctx.__exit__(*exc)
            self.compute.wait_stream(self.copy)
        return False


class PackedImages:
    """A batch of variable-size uint8 RGB images packed back to back (256-byte aligned) in one device buffer.

    Host images are copied once, into a pinned staging buffer, by a few memcpy threads inside one C call, then cross PCIe in ONE
    asynchronous transfer; Pillow images travel as their in-memory RGBX bytes and are repacked to RGB by mq_unpack_rgbx."""

    def __init__(self, images: Sequence[ArrayLike], device: torch.device, channels: int = 3, pil_sizes=None):
        """pil_sizes = (heights, widths) int32 arrays: `images` is a list of loaded Pillow RGB images, vouched for by `pil_rgb_sizes` — the
        whole pack then runs without a Python statement per image (the per-image container / isinstance / shape bookkeeping below is ~4 us
        per image under the GIL: 1 ms of a 256-image request)."""
        fast = pil_sizes is not None
        imgs = list(images) if fast else [_as_u8_hwc(i, channels) for i in images]   # channels = 4: RGBA sources of mq_resize_mode_u8 (no Rgbx views among them)
        self.n = len(imgs)
        if fast:
            self.heights, self.widths = np.ascontiguousarray(pil_sizes[0], dtype=np.int32), np.ascontiguousarray(pil_sizes[1], dtype=np.int32)
        else:
            self.heights = np.asarray([i.shape[0] for i in imgs], dtype=np.int32)
            self.widths = np.asarray([i.shape[1] for i in imgs], dtype=np.int32)
        npix = self.heights.astype(np.int64) * self.widths.astype(np.int64)
        sizes = npix * channels
        padded = _align256(sizes)
        self.offsets = np.zeros(self.n, dtype=np.int64)
        if self.n > 1:
            self.offsets[1:] = np.cumsum(padded)[:-1]
        total = int(padded.sum()) if self.n else 0
        if self.n and not fast and all(isinstance(i, torch.Tensor) and i.device.type == "cuda" for i in imgs):
            if len({tuple(i.shape) for i in imgs}) == 1 and int(sizes[0]) % 256 == 0:
                buf = torch.stack([i.to(device) for i in imgs]).reshape(-1)       # equal sizes: one kernel instead of one copy per image
            else:
                buf = torch.empty(total, dtype=torch.uint8, device=device)
                for i, o, sz in zip(imgs, self.offsets, sizes):
                    buf[int(o):int(o + sz)] = i.reshape(-1).to(device)
            self.buffer = buf
            return
        # ---- host staging: [RGB images at their final offsets | RGBX images | unpack job table] ----
        # (layout arithmetic vectorised: per-image Python under the GIL is what serialises concurrent request threads)
        is_x = np.ones(self.n, dtype=bool) if fast else np.fromiter((isinstance(i, Rgbx) for i in imgs), dtype=bool, count=self.n)
        nx = int(is_x.sum())
        x_sizes = np.where(is_x, _align256(npix * 4), 0)
        x_off = total + np.cumsum(x_sizes) - x_sizes          # int64 [n]; meaningful where is_x
        cur = total + int(x_sizes.sum())
        jobs_off = _align256(cur)
        stage_bytes = jobs_off + nx * 24
        host = torch.empty(max(stage_bytes, 1), dtype=torch.uint8, pin_memory=torch.cuda.is_available())
        hnp = host.numpy()

        # Pillow images that have not been exported yet go to the native stager in ONE call (Arrow export + memcpy threads, GIL released
        # for the copies); everything else is copied by ONE mq_host_gather call (same threads, GIL released once for the whole pack)
        stager = L.load_stage()
        if fast:
            lazy, eager = list(range(self.n)), []
        else:
            lazy = [k for k in np.flatnonzero(is_x).tolist() if imgs[k]._view is None] if (stager is not None and nx) else []
            lazy_set = set(lazy)
            eager = [k for k in range(self.n) if k not in lazy_set] if lazy else range(self.n)
        ne = len(eager)
        srcs, nbytes, dsts = (C.c_void_p * max(ne, 1))(), np.empty(ne, dtype=np.int64), np.empty(ne, dtype=np.int64)
        keep = []
        for e, k in enumerate(eager):
            i = imgs[k]
            if is_x[k]:
                a, dsts[e] = i.view, x_off[k]
            else:
                a = i if isinstance(i, np.ndarray) else (i.numpy() if i.device.type == "cpu" else i.cpu().numpy())
                dsts[e] = self.offsets[k]
            if not a.flags.c_contiguous:
                a = np.ascontiguousarray(a)
            keep.append(a)
            srcs[e] = a.__array_interface__["data"][0]
            nbytes[e] = a.nbytes
        lib = L.load()
        staged = None
        with _H2D(device) as h2d:
            if lazy:
                lz_off = np.ascontiguousarray(x_off[lazy])
                lz_len = np.ascontiguousarray(npix[lazy] * 4)
                # all-Pillow batch: the RGBX region [total, cur) fills front to back in request order -> ship it slice by slice
                sliced = PACK_SLICE and len(lazy) == self.n and self.n >= 2 * PACK_SLICE and torch.cuda.is_available()
                step = PACK_SLICE if sliced else len(lazy)
                if sliced:
                    staged = torch.empty(max(stage_bytes, 1), dtype=torch.uint8, device=device)
                for j0 in range(0, len(lazy), step):
                    j1 = min(j0 + step, len(lazy))
                    for f in stager.gather_rgbx(imgs[j0:j1] if fast else [imgs[k].image for k in lazy[j0:j1]], host.data_ptr(), host.numel(), lz_off[j0:j1],
                                                lz_len[j0:j1], PACK_THREADS):
                        k = lazy[j0 + f]   # Pillow could not export this one zero-copy (e.g. an image stored in several blocks): its .view copies
                        v = Rgbx(imgs[k]).view if fast else imgs[k].view
                        hnp[int(x_off[k]):int(x_off[k]) + v.nbytes] = np.ascontiguousarray(v).reshape(-1)
                    if sliced:
                        lo, hi = int(x_off[lazy[j0]]), (int(x_off[lazy[j1]]) if j1 < len(lazy) else cur)
                        staged[lo:hi].copy_(host[lo:hi], non_blocking=True)
            # Experiment knob: a batch of one kind (all plain RGB arrays, or all Pillow RGBX views) fills the staging buffer front to back, so it
            # CAN be packed and shipped in PACK_CHUNKS pieces, the pinned H2D copy of piece c running while the memcpy threads pack piece c + 1.
            pieces = PACK_CHUNKS if (not lazy and nx in (0, self.n) and self.n >= 16 * PACK_CHUNKS and torch.cuda.is_available()) else 1
            if pieces > 1:
                staged = torch.empty(max(stage_bytes, 1), dtype=torch.uint8, device=device)
                end = cur if nx else total
                for c in range(pieces):
                    k0, k1 = c * self.n // pieces, (c + 1) * self.n // pieces
                    sub = (C.c_void_p * (k1 - k0))(*srcs[k0:k1])
                    L.check(lib.mq_host_gather_checked(sub, nbytes[k0:k1].ctypes.data, dsts[k0:k1].ctypes.data, k1 - k0, host.data_ptr(),
                                                       host.numel(), PACK_THREADS), "mq_host_gather")
                    lo, hi = int(dsts[k0]), (int(dsts[k1]) if k1 < self.n else end)
                    staged[lo:hi].copy_(host[lo:hi], non_blocking=True)
            elif ne:
                L.check(lib.mq_host_gather_checked(srcs, nbytes.ctypes.data, dsts.ctypes.data, ne, host.data_ptr(), host.numel(), PACK_THREADS),
                        "mq_host_gather")
            del keep
            if nx:
                jobs = np.ascontiguousarray(np.stack([x_off[is_x], self.offsets[is_x], npix[is_x]], axis=1), dtype=np.int64)
                hnp[jobs_off:jobs_off + nx * 24] = jobs.view(np.uint8).reshape(-1)
                if pieces > 1 or staged is not None:
                    staged[jobs_off:jobs_off + nx * 24].copy_(host[jobs_off:jobs_off + nx * 24], non_blocking=True)
            if staged is None:
                staged = host.to(device, non_blocking=True)
            h2d.hand_over(staged)
        self._host = host   # (pinned source of the asynchronous copies: kept until the consumer's kernels are enqueued behind them)
        if not nx:
            self.buffer = staged[:total] if stage_bytes != total else staged
            return
        lib = L.load()
        with torch.cuda.device(device):
            if nx == self.n:   # nothing but RGBX: the packed buffer is a fresh allocation
                buf = torch.empty(max(total, 1), dtype=torch.uint8, device=device)
            else:              # mixed: the RGB images already sit at their offsets in the staged copy; unpack the others beside them
                buf = staged
            L.check(lib.mq_unpack_rgbx(staged.data_ptr(), jobs_off, nx, int(npix[is_x].max()), buf.data_ptr(),
                                       torch.cuda.current_stream(device).cuda_stream), "mq_unpack_rgbx")
        self._staged = staged
        self.buffer = buf


class ImagePreprocessor:
    """Owns the scratch workspace; thread-compatible (one instance per calling thread or externally locked)."""

    def __init__(self, device: str, image_size: int, mean: Sequence[float] = OPENAI_DATASET_MEAN,
                 std: Sequence[float] = OPENAI_DATASET_STD):
        if not str(device).startswith("cuda") or not torch.cuda.is_available():
            raise L.MarqoHipUnavailableError("image preprocessing runs on the GPU only; there is no CPU fallback in marqo_amd")
        self.device = torch.device(device)
        self.lib = L.load()
        self._ops = L.load_torch_ops() if L.boundary() == "torch_ops" else None
        self.S = int(image_size)
        self.mean = (C.c_float * 3)(*mean)
        self.std = (C.c_float * 3)(*std)
        self._ws_by_stream: dict = {}
        self._ws_lock = threading.Lock()

    def _workspace(self, nbytes: int) -> torch.Tensor:
        """scratch of the CURRENT stream.  The same preprocessor is called from several streams (`.preprocess` from the download threads'
        default stream, encode() from every request thread's private stream): each stream keeps its own block — allocated while that stream
        is current, so the caching allocator recycles it behind that stream's work — and a block is only ever replaced by a bigger one
        (one shared block + record_stream on the NEW stream protected nothing and re-allocated on every stream change)."""
        cur = torch.cuda.current_stream(self.device).cuda_stream
        with self._ws_lock:
            ws = self._ws_by_stream.get(cur)
            if ws is None or ws.numel() < nbytes:
                ws = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self.device)
                if len(self._ws_by_stream) >= 64:   # streams come and go with request threads: do not keep scratch of dead ones forever
                    self._ws_by_stream.clear()
                self._ws_by_stream[cur] = ws
            return ws

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _resize_by_mode(self, images: Sequence, out_h: int, out_w: int, filt: int, crop: bool) -> torch.Tensor:
        """a batch that holds palette / bilevel or translucent RGBA sources: every mode group goes through the call that resizes it as
        Pillow resizes that mode (mq_resize_mode_u8) and lands at its request positions"""
        kinds = [_kind(i) for i in images]
        with torch.cuda.device(self.device):
            out = torch.empty(len(images), out_h, out_w, 3, dtype=torch.uint8, device=self.device)
            keep = []
            for mode in sorted(set(kinds)):
                idx = [k for k, m in enumerate(kinds) if m == mode]
                group = [images[k] if mode == L.MQ_IMG_RGB else images[k].array for k in idx]
                p = PackedImages(group, self.device, channels=4 if mode == L.MQ_IMG_RGBA else 3)
                sub = out if len(idx) == len(images) else torch.empty(len(idx), out_h, out_w, 3, dtype=torch.uint8, device=self.device)
                need = self.lib.mq_resize_mode_workspace_bytes(p.heights.ctypes.data, p.widths.ctypes.data, p.n, out_h, out_w, filt,
                                                               1 if crop else 0, mode)
                # (one scratch buffer per group: the groups' kernels are all in flight on this stream before the first one has run)
                ws = torch.empty(int(need) + 256, dtype=torch.uint8, device=self.device)
                L.check(self.lib.mq_resize_mode_u8(p.buffer.data_ptr(), p.offsets.ctypes.data, p.heights.ctypes.data, p.widths.ctypes.data,
                                                   p.n, out_h, out_w, filt, 1 if crop else 0, mode, sub.data_ptr(), ws.data_ptr(), ws.numel(),
                                                   self._stream()), "mq_resize_mode_u8")
                if sub is not out:
                    out.index_copy_(0, torch.as_tensor(idx, dtype=torch.int64).to(self.device, non_blocking=True), sub)
                keep.append((p, ws, sub))
            self._keep = keep   # sources / scratch must outlive the enqueued kernels
        return out

    def resize_crop_u8(self, images: Sequence[ArrayLike], pil_sizes=None) -> torch.Tensor:
        """Resize(S, bicubic) + CenterCrop(S): list of uint8 [H_i, W_i, 3] (or pixel containers of other image modes, pil_pixels)
        -> uint8 [n, S, S, 3] on device.  pil_sizes: `images` are loaded Pillow RGB images (PackedImages' batch fast path)."""
        if pil_sizes is None and any(isinstance(i, (Rgba, NearestRgb)) for i in images):
            return self._resize_by_mode(images, self.S, self.S, 3, crop=True)
        with torch.cuda.device(self.device):
            p = PackedImages(images, self.device, pil_sizes=pil_sizes)
            out = torch.empty(p.n, self.S, self.S, 3, dtype=torch.uint8, device=self.device)
            if p.n == 0:
                return out
            need = self.lib.mq_clip_resize_workspace_bytes(p.heights.ctypes.data, p.widths.ctypes.data, p.n, self.S)
            ws = self._workspace(need)
            if self._ops is not None:   # the PyTorch custom-op face of the same entry point (csrc/torch_ops.cpp)
                self._ops.clip_resize_crop_u8(p.buffer, torch.from_numpy(p.offsets), torch.from_numpy(p.heights), torch.from_numpy(p.widths),
                                              self.S, out, ws)
            else:
                L.check(self.lib.mq_clip_resize_crop_u8(p.buffer.data_ptr(), p.offsets.ctypes.data, p.heights.ctypes.data,
                                                        p.widths.ctypes.data, p.n, self.S, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                        self._stream()), "mq_clip_resize_crop_u8")
            self._keep = p  # the packed source must outlive the enqueued kernels
        return out

    def resize_u8(self, images: Sequence[ArrayLike], out_h: int, out_w: int, interpolation: str = "bicubic", pil_sizes=None) -> torch.Tensor:
        """PIL.Image.resize((out_w, out_h), BICUBIC | BILINEAR) of every image -> uint8 [n, out_h, out_w, 3] on device."""
        filt = {"bicubic": 3, "bilinear": 2}[interpolation]   # Pillow's Image.BICUBIC / Image.BILINEAR
        if pil_sizes is None and any(isinstance(i, (Rgba, NearestRgb)) for i in images):
            return self._resize_by_mode(images, out_h, out_w, filt, crop=False)
        with torch.cuda.device(self.device):
            p = PackedImages(images, self.device, pil_sizes=pil_sizes)
            out = torch.empty(p.n, out_h, out_w, 3, dtype=torch.uint8, device=self.device)
            if p.n == 0:
                return out
            need = self.lib.mq_resize_filter_workspace_bytes(p.heights.ctypes.data, p.widths.ctypes.data, p.n, out_h, out_w, filt)
            ws = self._workspace(need)
            L.check(self.lib.mq_resize_filter_u8(p.buffer.data_ptr(), p.offsets.ctypes.data, p.heights.ctypes.data, p.widths.ctypes.data,
                                                 p.n, out_h, out_w, filt, out.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()),
                    "mq_resize_filter_u8")
            self._keep = p
        return out

    def chunk_grid_u8(self, images: Sequence[ArrayLike], hn: int = 3, wn: int = 3, overlap: bool = False
                      ) -> Tuple[torch.Tensor, np.ndarray]:
        """'simple' / 'overlap' patch methods: -> (uint8 [n*count, S, S, 3] on device, boxes float32 [n, count, 4])."""
        count = self.lib.mq_chunk_grid_count(hn, wn, 1 if overlap else 0)
        if count <= 0:
            raise ValueError(f"bad chunk grid hn={hn} wn={wn}")
        images = [flatten_pixels(i) for i in images]   # (the chunker keeps the flatten-first treatment of palette / translucent sources)
        with torch.cuda.device(self.device):
            p = PackedImages(images, self.device)
            out = torch.empty(p.n * count, self.S, self.S, 3, dtype=torch.uint8, device=self.device)
            boxes = np.zeros((p.n, count, 4), dtype=np.float32)
            if p.n == 0:
                return out, boxes
            need = self.lib.mq_chunk_grid_workspace_bytes(p.heights.ctypes.data, p.widths.ctypes.data, p.n, hn, wn,
                                                          1 if overlap else 0, self.S)
            ws = self._workspace(need)
            L.check(self.lib.mq_chunk_grid_u8(p.buffer.data_ptr(), p.offsets.ctypes.data, p.heights.ctypes.data,
                                              p.widths.ctypes.data, p.n, hn, wn, 1 if overlap else 0, self.S, out.data_ptr(),
                                              boxes.ctypes.data, ws.data_ptr(), ws.numel(), self._stream()), "mq_chunk_grid_u8")
            self._keep = p
        return out, boxes

    def to_tensor_normalize(self, u8: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """uint8 [n, S, S, 3] -> fp32 [n, 3, S, S] (ToTensor + Normalize); `out`: a contiguous fp32 [n, 3, S, S] destination on the device (a slice
        of the loaders' preprocess slab)"""
        if u8.dtype != torch.uint8 or u8.ndim != 4 or tuple(u8.shape[1:]) != (self.S, self.S, 3):
            raise ValueError(f"expected uint8 [n, {self.S}, {self.S}, 3], got {u8.dtype} {tuple(u8.shape)}")
        with torch.cuda.device(self.device):
            u8 = u8.to(self.device).contiguous()
            if out is None:
                out = torch.empty(u8.shape[0], 3, self.S, self.S, dtype=torch.float32, device=self.device)
            elif out.dtype != torch.float32 or tuple(out.shape) != (u8.shape[0], 3, self.S, self.S) or not out.is_contiguous() or out.device != u8.device:
                raise ValueError("to_tensor_normalize: `out` must be a contiguous fp32 [n, 3, S, S] tensor on the preprocessor's device")
            L.check(self.lib.mq_to_tensor_normalize(u8.data_ptr(), out.data_ptr(), u8.shape[0], self.S, self.mean, self.std,
                                                    self._stream()), "mq_to_tensor_normalize")
        return out
