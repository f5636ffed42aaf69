"""Device-side tokenisation (K14): hash-table / character-table builders + thin wrappers over mq_tokenize_wordpiece / mq_tokenize_clip_bpe.

The reference tokenises on the host (open_clip_model.py:277, hugging_face_model.py:179-185).  Here the host tokenisers of
``engine/tokenizers.py`` stay the definition of record; the same algorithms run on the GPU for ANY UTF-8 text (one thread per text
for the splitting, one per word for the vocabulary work) and produce IDENTICAL ids (tests/test_gpu_tokenizers.py: fixed cases + a
seeded multi-script fuzz).  Per-character behaviour — whitespace, punctuation, CJK isolation, control-character removal,
lower-casing, accent stripping, the regex classes \\s / \\p{L} / \\p{N} — comes from a 64-bit-per-code-point table built HERE from
Python's own `unicodedata` / `str.lower` / `regex`, i.e. from the very functions the host tokenisers call.  What the table cannot
express (context-dependent characters: Greek capital sigma's final form, combining marks under a cased vocabulary's NFC step, code
points beyond U+2FFFF, ...), special-token spellings and — for CLIP — `&` (html.unescape) send a text to the host tokeniser; its ids
are patched into the same matrix, so callers see one result regardless of the route.

The table builders are plain numpy (no GPU needed) so that the CPU test suite can feed the very same tables to the host
build of the product algorithm (oracle/tokenize_host.cpp).
"""
from __future__ import annotations

import ctypes as C
import functools
import os
import re
import threading
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from marqo_amd import _lib as L
from marqo_amd.engine.tokenizers import ClipBpeTokenizer, WordPieceTokenizer, _byte_to_unicode

WP_ENTRY = np.dtype([("hash", "<u8"), ("id", "<i4"), ("off_len", "<u4")])
BPE_ENTRY = np.dtype([("key", "<u4"), ("rank", "<u4"), ("merged", "<u4"), ("pad", "<u4")])
assert WP_ENTRY.itemsize == 16 and BPE_ENTRY.itemsize == 16

_FNV_OFFSET, _FNV_PRIME, _CONT_SEED, _M64 = 0xcbf29ce484222325, 0x100000001b3, 0x9e3779b97f4a7c15, (1 << 64) - 1
MAX_WORD_CHARS_DEVICE = 100
MAX_TEXT_BYTES_DEVICE = 1 << 20

# ---- Unicode character table (layout: csrc/tokenize_algo.h) ---------------------------------------------------------------------
UNI_LIMIT = 0x30000
U_DROP, U_WS, U_ISOLATE, U_HOST, U_LETTER, U_NUMBER, U_HANGUL, U_STRIP = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80


def _entry(flags: int, out: Sequence[int] = ()) -> int:
    e = flags | (len(out) << 8)
    if len(out) > 0:
        e |= out[0] << 10
    if len(out) > 1:
        e |= out[1] << 31
    return e


@functools.lru_cache(maxsize=4)
def build_unicode_table(kind: str, lower: bool) -> np.ndarray:
    """uint64 [UNI_LIMIT]: what the HOST tokeniser of `kind` ('wordpiece' | 'clip') does to each code point on its own, derived by
    calling the host tokeniser's own helpers, plus U_HOST where a code point's treatment depends on its neighbours:

      wordpiece (WordPieceTokenizer._basic): drop (U+0000, U+FFFD, category C*) | whitespace (\\t \\n \\r, Zs, anything str.split
        splits on) | CJK ideograph -> isolated | NFC | lower() + NFD + drop Mn (lower-casing vocabularies) | punctuation -> isolated.
        lower = True: a combining mark (Mn) vanishes whatever it follows, so it is a plain DROP; precomposed Hangul decomposes
        arithmetically (U_HANGUL); U+03A3 (final-sigma rule of str.lower) and non-Mn characters with a canonical combining class
        (reordered by NFD) are U_HOST.   lower = False: NFC is contextual -> every mark (M*), conjoining jamo vowel / trailing
        consonant and anything with a combining class is U_HOST.
      clip (ClipBpeTokenizer.encode): lower() then the regex classes \\s, \\p{L}, \\p{N} (from the `regex` module, as the host);
        U+03A3, U+0130 (two-character lowering) and U+017F (matches 's' under IGNORECASE in the contraction alternatives) are U_HOST.
    """
    import unicodedata
    import regex
    from marqo_amd.engine.tokenizers import _is_cjk, _is_control, _is_punct, _is_ws
    tab = np.zeros(UNI_LIMIT, dtype=np.uint64)
    re_s, re_l, re_n = regex.compile(r"\s"), regex.compile(r"\p{L}"), regex.compile(r"\p{N}")
    for cp in range(UNI_LIMIT):
        if 0xD800 <= cp <= 0xDFFF:
            tab[cp] = _entry(U_HOST)
            continue
        ch = chr(cp)
        if kind == "wordpiece":
            cat = unicodedata.category(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                tab[cp] = _entry(U_DROP)
            elif _is_ws(ch) or ch.isspace():
                tab[cp] = _entry(U_WS)
            elif _is_cjk(cp):
                out = unicodedata.normalize("NFD" if lower else "NFC", unicodedata.normalize("NFC", ch))
                tab[cp] = _entry(U_ISOLATE, [ord(out)]) if len(out) == 1 and not _is_punct(out) else _entry(U_HOST)
            elif lower:
                if cp == 0x03A3 or (unicodedata.combining(ch) != 0 and cat != "Mn"):
                    tab[cp] = _entry(U_HOST)
                    continue
                if 0xAC00 <= cp <= 0xD7A3:
                    tab[cp] = _entry(U_HANGUL)
                    continue
                w = unicodedata.normalize("NFC", ch).lower()
                out = [c for c in unicodedata.normalize("NFD", w) if unicodedata.category(c) != "Mn"]
                if not out:
                    tab[cp] = _entry(U_DROP)
                elif len(out) == 1 and _is_punct(out[0]):
                    tab[cp] = _entry(U_ISOLATE, [ord(out[0])])
                elif len(out) <= 2 and not any(_is_punct(c) or _is_ws(c) or c.isspace() or _is_cjk(ord(c)) or _is_control(c) for c in out):
                    tab[cp] = _entry(0, [ord(c) for c in out])
                else:
                    tab[cp] = _entry(U_HOST)
            else:
                if cat.startswith("M") or unicodedata.combining(ch) != 0 or 0x1161 <= cp <= 0x1175 or 0x11A8 <= cp <= 0x11C2:
                    tab[cp] = _entry(U_HOST)
                    continue
                out = unicodedata.normalize("NFC", ch)
                if len(out) == 1 and _is_punct(out):
                    tab[cp] = _entry(U_ISOLATE, [ord(out)])
                elif len(out) <= 2 and not any(_is_punct(c) or _is_ws(c) or c.isspace() or _is_cjk(ord(c)) or _is_control(c) for c in out):
                    tab[cp] = _entry(0, [ord(c) for c in out])
                else:
                    tab[cp] = _entry(U_HOST)
        elif kind == "clip":
            out = ch.lower() if lower else ch
            if cp in (0x03A3, 0x017F) or len(out) != 1:
                tab[cp] = _entry(U_HOST)
                continue
            flags = U_WS if re_s.match(out) else (U_LETTER if re_l.match(out) else (U_NUMBER if re_n.match(out) else 0))
            if ch.isspace() and not (flags & U_WS):   # _clean_text's str.strip() removes these at the ends of a text (U+001C..U+001F)
                flags |= U_STRIP
            tab[cp] = _entry(flags, [ord(out)])
        else:
            raise ValueError(kind)
    return tab


def _fnv(data: bytes, cont: bool) -> int:
    h = _FNV_OFFSET ^ _CONT_SEED if cont else _FNV_OFFSET
    for b in data:
        h = ((h ^ b) * _FNV_PRIME) & _M64
    return h


def _pow2_at_least(n: int) -> int:
    p = 1024
    while p < n:
        p *= 2
    return p


def build_wordpiece_table(tok: WordPieceTokenizer) -> Dict[str, object]:
    """vocabulary -> open-addressing table (layout: csrc/tokenize_algo.h mq_wp_entry), keyed by the UTF-8 bytes of every piece of at
    most 127 bytes ('##x...' entries are continuation pieces).  A vocabulary holding a longer piece cannot be matched exactly by the
    device path and is refused."""
    if tok.max_chars > MAX_WORD_CHARS_DEVICE:
        raise ValueError(f"max_chars_per_word {tok.max_chars} > {MAX_WORD_CHARS_DEVICE} is not supported on the device")
    items: List[Tuple[bytes, bool, int]] = []
    for s, idx in tok.vocab.items():
        cont = s.startswith("##") and len(s) > 2
        try:
            data = (s[2:] if cont else s).encode("utf-8")
        except UnicodeEncodeError:  # lone surrogates: such a piece can never equal a substring of well-formed text
            continue
        if len(data) > 127:
            raise ValueError(f"vocabulary piece of {len(data)} bytes is too long for the device WordPiece table")
        if len(data) > 0:
            items.append((data, cont, int(idx)))
    n_slots = _pow2_at_least(2 * len(items) + 1)
    slots = np.zeros(n_slots, dtype=WP_ENTRY)
    slots["id"] = -1
    pool = bytearray()
    mask = n_slots - 1
    for data, cont, idx in items:
        h = _fnv(data, cont)
        slot = ((h ^ (h >> 32)) & 0xffffffff) & mask
        while slots["id"][slot] >= 0:
            slot = (slot + 1) & mask
        slots[slot] = (h, idx, (len(pool) << 8) | (int(cont) << 7) | len(data))
        pool += data
    if len(pool) >= 1 << 24:
        raise ValueError("vocabulary string pool too large for the 24-bit offsets")
    return dict(slots=slots, pool=np.frombuffer(bytes(pool) + b"\0" * 16, dtype=np.uint8).copy(), n_slots=n_slots,
                unk_id=tok.unk_id, cls_id=tok.cls_id, sep_id=tok.sep_id, pad_id=tok.pad_id, lower=int(bool(tok.lower)),
                max_word_chars=int(tok.max_chars))


def build_bpe_table(tok: ClipBpeTokenizer) -> Dict[str, object]:
    """merges -> pair table (layout: mq_bpe_entry).  Symbols are vocabulary ids; python-dict semantics are kept for
    duplicated merges (the LAST rank wins) and duplicated vocabulary strings (the LAST id wins)."""
    if tok.vocab_size > 65535:
        raise ValueError("CLIP BPE device path needs vocabulary ids < 65536")
    enc = tok.encoder
    pairs: Dict[int, Tuple[int, int]] = {}
    for (x, y), r in tok.rank.items():
        pairs[(enc[x] << 16) | enc[y]] = (r, enc[x + y])
    n_slots = _pow2_at_least(2 * len(pairs) + 1)
    slots = np.zeros(n_slots, dtype=BPE_ENTRY)
    slots["key"] = 0xffffffff
    mask = n_slots - 1
    for key, (r, m) in pairs.items():
        slot = ((((key * 0x9e3779b1) & 0xffffffff) >> 7) & mask)
        while slots["key"][slot] != 0xffffffff:
            slot = (slot + 1) & mask
        slots[slot] = (key, r, m, 0)
    b2u = _byte_to_unicode()
    byte_id = np.array([enc[b2u[b]] for b in range(256)], dtype=np.uint16)
    byte_end_id = np.array([enc[b2u[b] + "</w>"] for b in range(256)], dtype=np.uint16)
    return dict(slots=slots, byte_id=byte_id, byte_end_id=byte_end_id, n_slots=n_slots, sot_id=tok.sot_id, eot_id=tok.eot_id,
                lower=int(bool(tok.lower)))


def pack_texts(texts: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
    """texts -> (UTF-8 uint8 blob, int64 byte offsets [n+1]).  Texts that cannot be encoded (lone surrogates) must have been routed to
    the host by the scope test."""
    if all(t.isascii() for t in texts):
        blob = "".join(texts).encode("ascii")
        lens = [len(t) for t in texts]
    else:
        enc = [t.encode("utf-8") for t in texts]
        blob = b"".join(enc)
        lens = [len(e) for e in enc]
    offsets = np.zeros(len(texts) + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    return np.frombuffer(bytearray(blob + b"\0"), dtype=np.uint8), offsets


def _encodable(text: str) -> bool:
    if text.isascii():
        return True
    try:
        text.encode("utf-8")
        return True
    except UnicodeEncodeError:
        return False


def wordpiece_in_scope(tok: WordPieceTokenizer, text: str) -> bool:
    """host-side routing test (the kernel flags character-level reasons itself): size, encodability, special-token spellings"""
    firsts = getattr(tok, "_special_firsts", None)
    if firsts is None:   # first characters of the special-token spellings: "[" for BERT vocabularies, "<" and "[" for MPNet's
        firsts = tok._special_firsts = "".join(sorted({s[0] for s in tok._specials if s}))
    return (len(text) < MAX_TEXT_BYTES_DEVICE // 4 and _encodable(text)
            and (not any(c in text for c in firsts) or not any(s in text for s in tok._specials)))


def clip_in_scope(tok: ClipBpeTokenizer, text: str) -> bool:
    if len(text) >= MAX_TEXT_BYTES_DEVICE // 4 or "&" in text or not _encodable(text):
        return False
    if "<" in text:
        low = text.lower()
        return tok.SOT not in low and tok.EOT not in low
    return True


# A lone short text — the search path's one query — is tokenised faster on the host (0.01-0.07 ms, word cache warm / cold) than by the device
# route's staging copy + three kernel launches + length D2H (0.12-0.15 ms; from 8 short texts on the device wins: 0.15 vs 0.4 ms;
# profiles/r02ak_tokenize_small.txt).  Ids are identical on both routes (tests/test_gpu_tokenizers.py), so the loaders pick per request.
HOST_TOKENIZE_MAX_CHARS = int(os.environ.get("MARQO_AMD_HOST_TOKENIZE_MAX_CHARS", "128"))


def prefers_host(texts: Sequence[str]) -> bool:
    return len(texts) == 1 and len(texts[0]) <= HOST_TOKENIZE_MAX_CHARS


class _DeviceTokenizerBase:
    def __init__(self, device: str):
        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise L.MarqoHipUnavailableError(f"device tokenisation needs a cuda device (got {device!r}); there is no CPU fallback here — "
                                             f"use the host tokenisers of engine/tokenizers.py")
        self.lib = L.load()
        self._keep: List[torch.Tensor] = []
        self._pin: Optional[torch.Tensor] = None   # reusable pinned staging buffer (allocating pinned memory per call costs ms)
        self._ws: Optional[torch.Tensor] = None
        self._lock = threading.Lock()              # staging buffer + workspace are per tokenizer: serialise concurrent callers

    def _up(self, arr: np.ndarray) -> torch.Tensor:
        t = torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1)).to(self.device)
        self._keep.append(t)
        return t

    def _split_routes(self, texts: Sequence[str], in_scope) -> List[int]:
        """indices of the texts that take the device route.  Fast path: ONE scope test over the concatenation (a batch is almost
        always all-ASCII); only when that fails are the texts tested one by one."""
        joined = "\n".join(texts)
        if len(joined) < MAX_TEXT_BYTES_DEVICE * 64 and in_scope(joined) and all(len(t) < MAX_TEXT_BYTES_DEVICE for t in texts):
            return list(range(len(texts)))
        return [i for i, t in enumerate(texts) if in_scope(t)]

    def _stage(self, texts: Sequence[str]):
        """-> (device blob uint8, device offsets int64, total bytes): one pinned staging buffer, two async H2D copies"""
        blob, offsets = pack_texts(texts)
        need = blob.nbytes + offsets.nbytes + 64
        if self._pin is None or self._pin.numel() < need:
            self._pin = torch.empty(int(need * 1.5) + 4096, dtype=torch.uint8).pin_memory()
        nb = (blob.nbytes + 15) // 16 * 16
        pin = self._pin.numpy()
        pin[:blob.nbytes] = blob
        pin[nb:nb + offsets.nbytes] = offsets.view(np.uint8)
        d = self._pin[:nb + offsets.nbytes].to(self.device, non_blocking=True)
        return d[:blob.nbytes], d[nb:].view(torch.int64), int(offsets[-1])

    def _workspace(self, n: int, total_bytes: int, cap: int) -> torch.Tensor:
        need = int(self.lib.mq_tokenize_workspace_bytes(n, total_bytes, cap))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(int(need * 1.25) + 256, dtype=torch.uint8, device=self.device)
        return self._ws


class DeviceWordPieceTokenizer(_DeviceTokenizerBase):
    """BERT WordPiece on the GPU for in-scope texts; `host` (a WordPieceTokenizer) handles the rest and defines the result."""

    def __init__(self, host: WordPieceTokenizer, device: str):
        super().__init__(device)
        self.host = host
        t = build_wordpiece_table(host)
        self.vocab = L.WordPieceVocab(d_slots=self._up(t["slots"]).data_ptr(), d_pool=self._up(t["pool"]).data_ptr(), n_slots=t["n_slots"],
                                      unk_id=t["unk_id"], cls_id=t["cls_id"], sep_id=t["sep_id"], pad_id=t["pad_id"], lower=t["lower"],
                                      max_word_chars=t["max_word_chars"],
                                      d_unicode=self._up(build_unicode_table("wordpiece", bool(host.lower))).data_ptr())

    def encode_device(self, texts: Sequence[str], max_length: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (ids int32 [n, max_length] on device, right-padded with pad_id; lengths int64 [n] on host)"""
        n = len(texts)
        ids = torch.full((n, max_length), self.host.pad_id, dtype=torch.int32, device=self.device)
        lens_np = np.zeros(n, dtype=np.int64)   # (host bookkeeping in NumPy: see engine/towers.py::_host_i64)
        lens = torch.from_numpy(lens_np)
        if n == 0:
            return ids, lens
        on_dev = self._split_routes(texts, lambda t: wordpiece_in_scope(self.host, t))
        dev_set = set(on_dev)
        if on_dev:
            sel = texts if len(on_dev) == n else [texts[i] for i in on_dev]
            m = len(sel)
            with self._lock, torch.cuda.device(self.device):
                d_blob, d_off, total = self._stage(sel)
                d_ids = ids if m == n else torch.empty(m, max_length, dtype=torch.int32, device=self.device)
                d_meta = torch.empty(2, m, dtype=torch.int32, device=self.device)
                ws = self._workspace(m, total, max(max_length - 2, 1))
                L.check(self.lib.mq_tokenize_wordpiece(C.byref(self.vocab), d_blob.data_ptr(), d_off.data_ptr(), m, total, max_length,
                                                       d_ids.data_ptr(), max_length, d_meta[0].data_ptr(), d_meta[1].data_ptr(), ws.data_ptr(),
                                                       ws.numel(), torch.cuda.current_stream(self.device).cuda_stream), "mq_tokenize_wordpiece")
                meta = d_meta.cpu().numpy()  # (also the sync after which the pinned staging buffer may be reused)
            if meta[1].any():  # defensive: the kernel disagreed with the host-side scope test
                bad = [on_dev[j] for j in np.nonzero(meta[1])[0].tolist()]
                dev_set.difference_update(bad)
            lens_np[on_dev] = meta[0]
            if m != n:
                ids[torch.tensor(on_dev, dtype=torch.int64).to(self.device)] = d_ids
        rest = [i for i in range(n) if i not in dev_set]
        if rest:
            enc = [self.host.encode(texts[i], max_length) for i in rest]
            block = np.full((len(rest), max_length), self.host.pad_id, dtype=np.int32)
            for j, e in enumerate(enc):
                block[j, :len(e)] = e
                lens_np[rest[j]] = len(e)
            ids[torch.tensor(rest, device=self.device)] = torch.from_numpy(block).to(self.device)
        return ids, lens

    def __call__(self, texts, max_length: Optional[int] = None) -> Dict[str, np.ndarray]:
        """drop-in for WordPieceTokenizer.__call__ (padding to the longest, truncation)"""
        if isinstance(texts, str):
            texts = [texts]
        cap = max_length if max_length is not None else 2 + max((len(t) for t in texts), default=0)
        ids, lens = self.encode_device(texts, cap)
        lens = lens.numpy()
        S = int(lens.max()) if len(texts) else 0
        out = ids[:, :S].cpu().numpy().astype(np.int64)
        mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int64)
        return {"input_ids": out, "attention_mask": mask}


class DeviceClipBpeTokenizer(_DeviceTokenizerBase):
    """CLIP byte-level BPE on the GPU for in-scope texts; `host` (a ClipBpeTokenizer) handles the rest."""

    def __init__(self, host: ClipBpeTokenizer, device: str):
        super().__init__(device)
        self.host = host
        t = build_bpe_table(host)
        self.vocab = L.ClipBpeVocab(d_slots=self._up(t["slots"]).data_ptr(), d_byte_id=self._up(t["byte_id"]).data_ptr(),
                                    d_byte_end_id=self._up(t["byte_end_id"]).data_ptr(), n_slots=t["n_slots"], sot_id=t["sot_id"],
                                    eot_id=t["eot_id"], lower=t["lower"],
                                    d_unicode=self._up(build_unicode_table("clip", bool(host.lower))).data_ptr())

    def encode_device(self, texts: Sequence[str], context_length: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (ids int32 [n, ctx] on device, zero padded; lengths int64 [n] on host = SOT..EOT)"""
        ctx = context_length or self.host.context_length
        n = len(texts)
        ids = torch.zeros(n, ctx, dtype=torch.int32, device=self.device)
        lens_np = np.zeros(n, dtype=np.int64)   # (host bookkeeping in NumPy: see engine/towers.py::_host_i64)
        lens = torch.from_numpy(lens_np)
        if n == 0:
            return ids, lens
        on_dev = self._split_routes(texts, lambda t: clip_in_scope(self.host, t))
        dev_set = set(on_dev)
        if on_dev:
            sel = texts if len(on_dev) == n else [texts[i] for i in on_dev]
            m = len(sel)
            with self._lock, torch.cuda.device(self.device):
                d_blob, d_off, total = self._stage(sel)
                d_ids = ids if m == n else torch.empty(m, ctx, dtype=torch.int32, device=self.device)
                d_meta = torch.empty(2, m, dtype=torch.int32, device=self.device)
                ws = self._workspace(m, total, ctx)
                L.check(self.lib.mq_tokenize_clip_bpe(C.byref(self.vocab), d_blob.data_ptr(), d_off.data_ptr(), m, total, ctx, d_ids.data_ptr(),
                                                      d_meta[0].data_ptr(), d_meta[1].data_ptr(), ws.data_ptr(), ws.numel(),
                                                      torch.cuda.current_stream(self.device).cuda_stream), "mq_tokenize_clip_bpe")
                meta = d_meta.cpu().numpy()  # (also the sync after which the pinned staging buffer may be reused)
            if meta[1].any():  # e.g. a pre-token longer than the device scratch
                bad = [on_dev[j] for j in np.nonzero(meta[1])[0].tolist()]
                dev_set.difference_update(bad)
            lens_np[on_dev] = meta[0]
            if m != n:
                ids[torch.tensor(on_dev, dtype=torch.int64).to(self.device)] = d_ids
        rest = [i for i in range(n) if i not in dev_set]
        if rest:
            block = np.ascontiguousarray(self.host([texts[i] for i in rest], ctx), dtype=np.int32)
            lens_np[rest] = block.argmax(axis=1) + 1
            ids[torch.tensor(rest, device=self.device)] = torch.from_numpy(block).to(self.device)
        return ids, lens

    def __call__(self, texts, context_length: Optional[int] = None) -> np.ndarray:
        """drop-in for ClipBpeTokenizer.__call__: int64 [n, ctx] on the host"""
        if isinstance(texts, str):
            texts = [texts]
        ids, _ = self.encode_device(texts, context_length)
        return ids.cpu().numpy().astype(np.int64)


# =====================================================================================================================================
# SentencePiece unigram on the device (XLM-RoBERTa: multilingual-e5; T5-style vocabularies: SigLIP)
# =====================================================================================================================================
SP_ENTRY = np.dtype([("hash", "<u8"), ("id", "<i4"), ("off_len", "<u4")])
SP_HOST = 0xFF


def _fnv_columns(mat: np.ndarray, lens: np.ndarray):
    """rolling FNV-1a-64 over the rows of a zero-padded uint8 matrix: yields (k, hash after k + 1 bytes) for every column"""
    h = np.full(mat.shape[0], _FNV_OFFSET, dtype=np.uint64)
    prime = np.uint64(_FNV_PRIME)
    with np.errstate(over="ignore"):
        for k in range(mat.shape[1]):
            live = lens > k
            h = np.where(live, (h ^ mat[:, k].astype(np.uint64)) * prime, h)
            yield k, h


def _insert_open_addressing(home: np.ndarray, n_slots: int) -> np.ndarray:
    """linear-probing placement of keys with the given home slots -> slot index of every key (vectorised: every pass places, among the
    keys that want the same free slot, the lowest-numbered one and moves the others on by one — exactly what sequential insertion in
    key order guarantees: no free slot between a key's home and its place)"""
    mask = n_slots - 1
    pos = home.astype(np.int64) & mask
    used = np.zeros(n_slots, dtype=bool)
    place = np.full(home.shape[0], -1, dtype=np.int64)
    pending = np.arange(home.shape[0], dtype=np.int64)
    while pending.size:
        p = pos[pending]
        free = ~used[p]
        cand, cp = pending[free], p[free]
        if cand.size:
            uniq, first = np.unique(cp, return_index=True)
            winners = cand[first]
            used[uniq] = True
            place[winners] = uniq
        pending = pending[place[pending] < 0]
        pos[pending] = (pos[pending] + 1) & mask
    return place


def build_sentencepiece_table(sp) -> Dict[str, object]:
    """SentencePieceProcessor (unigram) -> the device tables of csrc/tokenize_algo.h::mq_sp_table:
      * hash table over every NORMAL piece AND every proper prefix (at character boundaries) of one, keyed by FNV-1a-64 of the bytes;
      * scores[id];
      * per-code-point normalisation map, read off the model's OWN normaliser: X(c) = Normalize('a' + c + 'b') minus the frame, checked in
        a second context; characters whose mapping depends on neighbours (marks, conjoining jamo — the NFKC-based map composes them),
        U+2581 itself and mappings that grow more than 3x are flagged for the host."""
    import unicodedata
    from sentencepiece import sentencepiece_model_pb2 as pb
    m = pb.ModelProto.FromString(sp.serialized_model_proto())
    if m.trainer_spec.model_type != 1:
        raise ValueError("device SentencePiece supports unigram models only")
    if m.trainer_spec.byte_fallback or any(p.type in (4, 6) for p in m.pieces):
        raise ValueError("device SentencePiece does not support byte-fallback / user-defined symbols")
    if m.trainer_spec.treat_whitespace_as_suffix or not m.normalizer_spec.escape_whitespaces:
        raise ValueError("device SentencePiece expects escaped whitespace as a prefix")
    unk_id = int(m.trainer_spec.unk_id)
    normal = [(i, p.piece.encode("utf-8"), float(p.score)) for i, p in enumerate(m.pieces) if p.type == 1]
    if not normal or max(len(b) for _, b, _ in normal) > 255:
        raise ValueError("vocabulary without pieces / with a piece longer than 255 bytes")
    scores = np.zeros(len(m.pieces), dtype=np.float32)
    for i, _, sc in normal:
        scores[i] = sc
    min_score = float(np.float32(min(sc for _, _, sc in normal)))
    # ---- pieces + prefixes -> keys
    ids = np.array([i for i, _, _ in normal], dtype=np.int64)
    lens = np.array([len(b) for _, b, _ in normal], dtype=np.int64)
    max_len = int(lens.max())
    mat = np.zeros((len(normal), max_len), dtype=np.uint8)
    pool = bytearray()
    starts = np.zeros(len(normal), dtype=np.int64)
    for r, (_, b, _) in enumerate(normal):
        mat[r, :len(b)] = np.frombuffer(b, dtype=np.uint8)
        starts[r] = len(pool)
        pool += b
    key_hash, key_len, key_row, key_full = [], [], [], []
    nxt = np.concatenate([mat[:, 1:], np.zeros((len(normal), 1), np.uint8)], axis=1)
    for k, h in _fnv_columns(mat, lens):
        boundary = (lens > k) & ((lens == k + 1) | ((nxt[:, k] & 0xC0) != 0x80))      # a character ends at byte k
        rows = np.nonzero(boundary)[0]
        key_hash.append(h[rows])
        key_len.append(np.full(rows.size, k + 1, dtype=np.int64))
        key_row.append(rows)
        key_full.append(lens[rows] == k + 1)
    key_hash, key_len, key_row, key_full = (np.concatenate(x) for x in (key_hash, key_len, key_row, key_full))
    # one entry per distinct (hash, length); a full piece wins over a mere prefix.  (64-bit collisions between DIFFERENT strings of one
    # length would merge two keys: the lookup verifies bytes, so a collision could only hide a key — checked below.)
    order = np.lexsort((~key_full, key_len, key_hash))
    kh, kl, kr, kf = key_hash[order], key_len[order], key_row[order], key_full[order]
    first = np.ones(kh.size, dtype=bool)
    first[1:] = (kh[1:] != kh[:-1]) | (kl[1:] != kl[:-1])
    kh, kl, kr, kf = kh[first], kl[first], kr[first], kf[first]
    n_keys = int(kh.size)
    n_slots = _pow2_at_least(2 * n_keys + 1)
    place = _insert_open_addressing((kh ^ (kh >> np.uint64(32))) & np.uint64(0xffffffff), n_slots)
    slots = np.zeros(n_slots, dtype=SP_ENTRY)
    slots["id"] = -1
    slots["hash"][place] = kh
    slots["id"][place] = np.where(kf, ids[kr], -2).astype(np.int32)
    slots["off_len"][place] = ((starts[kr] << 8) | kl).astype(np.uint32)
    distinct = {bytes(b[:k]) for _, b, _ in normal for k in range(1, len(b) + 1) if k == len(b) or (b[k] & 0xC0) != 0x80}
    if len(distinct) != n_keys:
        raise ValueError("FNV-1a-64 collision inside the SentencePiece vocabulary")
    # ---- per-code-point normalisation map
    nmap = np.zeros(UNI_LIMIT, dtype=np.uint32)
    npool = bytearray()
    cache: Dict[str, int] = {}
    host = SP_HOST
    norm = sp.normalize
    frame = norm("ab")
    pre, suf = frame[:-1], "b"
    if frame != pre + suf or not norm("a b").startswith(pre):
        raise ValueError("unexpected normaliser framing")
    frame2 = norm("かア")
    composing = set()
    for c in range(0x110000):
        d = unicodedata.decomposition(chr(c))
        if d and not d.startswith("<"):
            parts = d.split()
            # (composition exclusions — Hebrew points, Devanagari nukta ... — decompose but never re-compose: they stay context-free)
            if len(parts) == 2 and len(unicodedata.normalize("NFC", chr(int(parts[0], 16)) + chr(int(parts[1], 16)))) == 1:
                composing.add(int(parts[1], 16))
    ccc = np.zeros(UNI_LIMIT, dtype=np.uint8)
    for cp in range(UNI_LIMIT):
        if 0xD800 <= cp <= 0xDFFF or cp == 0x2581:
            nmap[cp] = host
            continue
        ch = chr(cp)
        # contextual under the NFKC-based map: a character that is the SECOND half of a canonical composition (it merges with the
        # character before it), conjoining jamo vowels / trailing consonants (algorithmic composition) -> host.  Other marks pass
        # through on their own; their canonical REORDERING (two adjacent marks with descending combining classes) is caught on the
        # device from the ccc table
        if cp in composing or 0x1160 <= cp <= 0x11FF or 0xD7B0 <= cp <= 0xD7FF:
            nmap[cp] = host
            continue
        ccc[cp] = unicodedata.combining(ch)
        r = norm("a" + ch + "b")
        if not (r.startswith(pre) and r.endswith(suf)):
            nmap[cp] = host
            continue
        x = r[len(pre):len(r) - 1]
        r2 = norm("か" + ch + "ア")
        if not (r2.startswith(frame2[:-1]) and r2.endswith(frame2[-1])) or r2[len(frame2) - 1:len(r2) - 1] != x:
            nmap[cp] = host
            continue
        x = x.replace("▁", " ")
        xb = x.encode("utf-8")
        if len(xb) > 3 * len(ch.encode("utf-8")) or len(xb) >= host:
            nmap[cp] = host
            continue
        off = cache.get(x)
        if off is None:
            off = cache[x] = len(npool)
            npool += xb
        nmap[cp] = (off << 8) | len(xb)
    return dict(slots=slots, pool=np.frombuffer(bytes(pool) + b"\0" * 16, dtype=np.uint8).copy(), scores=scores, nmap=nmap,
                npool=np.frombuffer(bytes(npool) + b"\0" * 16, dtype=np.uint8).copy(), ccc=ccc, n_slots=n_slots, unk_id=unk_id,
                unk_score=float(np.float32(min_score) - np.float32(10.0)), add_dummy_prefix=int(m.normalizer_spec.add_dummy_prefix),
                remove_extra_ws=int(m.normalizer_spec.remove_extra_whitespaces), max_piece_bytes=max_len)


class DeviceSentencePieceTokenizer(_DeviceTokenizerBase):
    """SentencePiece unigram on the GPU.  `host` is the host tokeniser of record: an XlmRobertaTokenizer (rows <s> ids + 1 ... </s>, <pad>
    padding, <unk> = 3) or a SiglipTokenizer backed by `spiece.model` (canonicalize on the host — three string operations — then rows
    ids ... </s> padded with </s> to the context length).  Texts the kernel flags (composing marks, conjoining jamo, reordering mark
    sequences, code points beyond U+2FFFF) are tokenised by `host` and patched in."""

    def __init__(self, host, device: str):
        super().__init__(device)
        self.host = host
        from marqo_amd.engine.tokenizers import SiglipTokenizer, XlmRobertaTokenizer
        if isinstance(host, XlmRobertaTokenizer):
            sp, frame = host.sp, dict(prefix_id=host.cls_id, suffix_id=host.sep_id, pad_id=host.pad_id, id_offset=host.FAIRSEQ_OFFSET, unk_out=host.unk_id)
            self.kind = "xlmr"
        elif isinstance(host, SiglipTokenizer) and host._sp is not None:
            sp, frame = host._sp, dict(prefix_id=-1, suffix_id=host.eos_id, pad_id=host.pad_id, id_offset=0, unk_out=int(host._sp.unk_id()))
            self.kind = "siglip"
        else:
            raise ValueError("DeviceSentencePieceTokenizer needs an XlmRobertaTokenizer or a sentencepiece-backed SiglipTokenizer")
        t = build_sentencepiece_table(sp)
        self.pad_id = frame["pad_id"]
        self.vocab = L.SentencePieceVocab(
            d_slots=self._up(t["slots"]).data_ptr(), d_pool=self._up(t["pool"]).data_ptr(), d_score=self._up(t["scores"]).data_ptr(),
            d_nmap=self._up(t["nmap"]).data_ptr(), d_npool=self._up(t["npool"]).data_ptr(), d_ccc=self._up(t["ccc"]).data_ptr(),
            n_slots=t["n_slots"], unk_id=t["unk_id"], unk_score=t["unk_score"], add_dummy_prefix=t["add_dummy_prefix"],
            remove_extra_ws=t["remove_extra_ws"], max_piece_bytes=t["max_piece_bytes"], **frame)

    def encode_device(self, texts: Sequence[str], max_length: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (ids int32 [n, max_length] on device, padded with the pad id; lengths int64 [n] on host, specials included)"""
        n = len(texts)
        ids = torch.full((n, max_length), self.pad_id, dtype=torch.int32, device=self.device)
        lens_np = np.zeros(n, dtype=np.int64)   # (host bookkeeping in NumPy: see engine/towers.py::_host_i64)
        lens = torch.from_numpy(lens_np)
        if n == 0:
            return ids, lens
        if self.kind == "siglip":
            from marqo_amd.engine.tokenizers import canonicalize_text
            texts = [canonicalize_text(t) for t in texts]
        on_dev = [i for i, t in enumerate(texts) if len(t) < MAX_TEXT_BYTES_DEVICE // 4 and _encodable(t)]
        dev_set = set(on_dev)
        if on_dev:
            sel = texts if len(on_dev) == n else [texts[i] for i in on_dev]
            m = len(sel)
            with self._lock, torch.cuda.device(self.device):
                d_blob, d_off, total = self._stage(sel)
                d_ids = ids if m == n else torch.empty(m, max_length, dtype=torch.int32, device=self.device)
                d_meta = torch.empty(2, m, dtype=torch.int32, device=self.device)
                need = int(self.lib.mq_tokenize_sentencepiece_workspace_bytes(m, total, max_length))
                if self._ws is None or self._ws.numel() < need:
                    self._ws = torch.empty(int(need * 1.25) + 256, dtype=torch.uint8, device=self.device)
                L.check(self.lib.mq_tokenize_sentencepiece(C.byref(self.vocab), d_blob.data_ptr(), d_off.data_ptr(), m, total, max_length,
                                                           d_ids.data_ptr(), max_length, d_meta[0].data_ptr(), d_meta[1].data_ptr(),
                                                           self._ws.data_ptr(), self._ws.numel(),
                                                           torch.cuda.current_stream(self.device).cuda_stream), "mq_tokenize_sentencepiece")
                meta = d_meta.cpu().numpy()
            if meta[1].any():
                dev_set.difference_update(on_dev[j] for j in np.nonzero(meta[1])[0].tolist())
            lens_np[on_dev] = meta[0]
            if m != n:
                ids[torch.tensor(on_dev, dtype=torch.int64).to(self.device)] = d_ids
        rest = [i for i in range(n) if i not in dev_set]
        if rest:
            block = np.full((len(rest), max_length), self.pad_id, dtype=np.int32)
            for j, i in enumerate(rest):
                if self.kind == "xlmr":
                    e = self.host.encode(texts[i], max_length)
                else:  # (already canonicalised above)
                    e = (list(self.host._sp.encode(texts[i]))[: max_length - 1]) + [self.host.eos_id]
                block[j, :len(e)] = e
                lens_np[i] = len(e)
            ids[torch.tensor(rest, device=self.device)] = torch.from_numpy(block).to(self.device)
        return ids, lens

    def __call__(self, texts, max_length: Optional[int] = None):
        """drop-in for the host tokeniser's __call__ (XLM-R: dict padded to the longest; SigLIP: int64 [n, ctx])"""
        if isinstance(texts, str):
            texts = [texts]
        if self.kind == "siglip":
            ids, _ = self.encode_device(texts, self.host.context_length)
            return ids.cpu().numpy().astype(np.int64)
        cap = max_length if max_length is not None else 2 + 4 * max((len(t) for t in texts), default=0)
        ids, lens = self.encode_device(texts, cap)
        lens = lens.numpy()
        S = int(lens.max()) if len(texts) else 0
        out = ids[:, :S].cpu().numpy().astype(np.int64)
        mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int64)
        return {"input_ids": out, "attention_mask": mask}
