"""Host-side tokenisers for the two text towers.

The reference tokenises on the host with third-party code: ``open_clip.get_tokenizer(...)`` (byte-level BPE,
open_clip_model.py:203-222, called at :277) and ``transformers.AutoTokenizer`` (WordPiece for the BERT family,
hugging_face_model.py:105-170, called at :179-185 with padding=True, truncation=True, max_length=tokens).
Neither open_clip nor any vocabulary file is available offline, so both published algorithms are implemented
here from their specification and read the standard vocabulary files from disk:

  * ``ClipBpeTokenizer``   — CLIP's ``bpe_simple_vocab_16e6.txt.gz`` merges file, context 77, SOT/EOT, zero pad
  * ``WordPieceTokenizer`` — BERT ``vocab.txt`` (or the ``model.vocab`` of a ``tokenizer.json``), basic
    tokenisation (clean, CJK spacing, lower-case + accent stripping, punctuation split) + greedy WordPiece

tests/test_tokenizers.py pins both against the independent implementations in `transformers` on synthetic
vocabularies.  ``SyntheticTokenizer`` is the stand-in used with random-init weights (no vocabulary on disk):
it is NOT a model tokenizer, only a deterministic text -> ids map so that the text path can be benchmarked.
"""
from __future__ import annotations

import gzip
import hashlib
import html
import json
import os
import unicodedata
from functools import lru_cache
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np
import regex as re

try:  # the reference's basic_clean runs ftfy.fix_text first (hf_tokenizer.py:13-17); optional here
    import ftfy as _ftfy
except Exception:  # pragma: no cover - ftfy is not in the image
    _ftfy = None


# ---------------------------------------------------------------------------------------------------
# CLIP byte-level BPE
# ---------------------------------------------------------------------------------------------------
@lru_cache()
def _byte_to_unicode() -> Dict[int, str]:
    """GPT-2 style reversible byte -> printable unicode map (printable latin-1 bytes map to themselves, the
    rest to code points from 256 up)."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    table, extra = {}, 0
    for b in keep:
        table[b] = chr(b)
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    # the vocabulary order is: kept bytes in `keep` order, then the remapped ones in byte order
    return {b: table[b] for b in keep + [b for b in range(256) if b not in keep]}


def _clean_text(text: str) -> str:
    if _ftfy is not None:
        text = _ftfy.fix_text(text)
    text = html.unescape(html.unescape(text)).strip()
    return re.sub(r"\s+", " ", text).strip()


class ClipBpeTokenizer:
    SOT, EOT = "<start_of_text>", "<end_of_text>"
    N_MERGES = 49152 - 256 - 2

    def __init__(self, merges: Union[str, Sequence[Tuple[str, str]]], context_length: int = 77, lower: bool = True):
        if isinstance(merges, str):
            opener = gzip.open if merges.endswith(".gz") else open
            with opener(merges, "rb") as f:
                lines = f.read().decode("utf-8").split("\n")
            merges = [tuple(l.split()) for l in lines[1:self.N_MERGES + 1] if l.strip()]  # line 0 is a version header
        self.merges: List[Tuple[str, str]] = [tuple(m) for m in merges]
        units = list(_byte_to_unicode().values())
        vocab = units + [u + "</w>" for u in units] + ["".join(m) for m in self.merges] + [self.SOT, self.EOT]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.rank = {m: i for i, m in enumerate(self.merges)}
        self.context_length = context_length
        self.lower = lower
        self.sot_id, self.eot_id = self.encoder[self.SOT], self.encoder[self.EOT]
        self._cache: Dict[str, List[int]] = {}
        self._pat = re.compile(
            re.escape(self.SOT) + "|" + re.escape(self.EOT) + r"""|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
            re.IGNORECASE)

    @property
    def vocab_size(self) -> int:
        return len(self.encoder)

    def _bpe_ids(self, token: str) -> List[int]:
        hit = self._cache.get(token)
        if hit is not None:
            return hit
        if token in (self.SOT, self.EOT):
            ids = [self.encoder[token]]
        else:
            word = [_byte_to_unicode()[b] for b in token.encode("utf-8")]
            word[-1] += "</w>"
            while len(word) > 1:
                best, best_rank = None, None
                for pair in zip(word[:-1], word[1:]):
                    r = self.rank.get(pair)
                    if r is not None and (best_rank is None or r < best_rank):
                        best, best_rank = pair, r
                if best is None:
                    break
                merged, i = [], 0
                while i < len(word):
                    if i + 1 < len(word) and word[i] == best[0] and word[i + 1] == best[1]:
                        merged.append(word[i] + word[i + 1])
                        i += 2
                    else:
                        merged.append(word[i])
                        i += 1
                word = merged
            ids = [self.encoder[w] for w in word]
        self._cache[token] = ids
        return ids

    def encode(self, text: str) -> List[int]:
        text = _clean_text(text)
        if self.lower:
            text = text.lower()
        out: List[int] = []
        for tok in self._pat.findall(text):
            out.extend(self._bpe_ids(tok))
        return out

    def __call__(self, texts: Union[str, Sequence[str]], context_length: Optional[int] = None) -> np.ndarray:
        """-> int64 [n, context_length]: SOT ids... EOT, zero-padded; over-long inputs are truncated and the last
        kept position overwritten with EOT (open_clip behaviour)."""
        if isinstance(texts, str):
            texts = [texts]
        L = context_length or self.context_length
        out = np.zeros((len(texts), L), dtype=np.int64)
        for i, t in enumerate(texts):
            ids = [self.sot_id] + self.encode(t) + [self.eot_id]
            if len(ids) > L:
                ids = ids[:L]
                ids[-1] = self.eot_id
            out[i, :len(ids)] = ids
        return out


# ---------------------------------------------------------------------------------------------------
# BERT WordPiece
# ---------------------------------------------------------------------------------------------------
def _is_punct(ch: str) -> bool:
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


def _is_ws(ch: str) -> bool:
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch: str) -> bool:
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


_ASCII_DROP_CONTROLS = {cp: None for cp in range(128) if (cp < 32 or cp == 127) and cp not in (9, 10, 13)}
_ASCII_BASIC = re.compile(r"[!-/:-@\[-`{-~]|[^ \t\n\r!-/:-@\[-`{-~]+")


class WordPieceTokenizer:
    def __init__(self, vocab: Union[str, Dict[str, int], Iterable[str]], do_lower_case: bool = True,
                 unk="[UNK]", cls="[CLS]", sep="[SEP]", pad="[PAD]", mask="[MASK]", max_chars_per_word: int = 100):
        if isinstance(vocab, str):
            vocab = self._read_vocab(vocab)
        elif not isinstance(vocab, dict):
            vocab = {tok: i for i, tok in enumerate(vocab)}
        self.vocab: Dict[str, int] = dict(vocab)
        self.lower = do_lower_case
        self.unk, self.cls, self.sep, self.pad, self.mask = unk, cls, sep, pad, mask
        for t in (unk, cls, sep, pad):
            if t not in self.vocab:
                raise ValueError(f"vocabulary has no {t} token")
        self.unk_id, self.cls_id, self.sep_id, self.pad_id = (self.vocab[t] for t in (unk, cls, sep, pad))
        self.max_chars = max_chars_per_word
        specials = [t for t in (unk, cls, sep, pad, mask) if t in self.vocab]
        self._special_split = re.compile("(" + "|".join(re.escape(t) for t in specials) + ")")
        self._specials = set(specials)
        self._cache: Dict[str, List[int]] = {}

    @staticmethod
    def _read_vocab(path: str) -> Dict[str, int]:
        if os.path.isdir(path):
            for name in ("vocab.txt", "tokenizer.json"):
                if os.path.isfile(os.path.join(path, name)):
                    path = os.path.join(path, name)
                    break
            else:
                raise FileNotFoundError(f"no vocab.txt / tokenizer.json under {path}")
        if path.endswith(".json"):
            with open(path, encoding="utf-8") as f:
                tj = json.load(f)
            model = tj.get("model", {})
            if model.get("type") != "WordPiece":
                raise ValueError(f"{path}: tokenizer model type {model.get('type')!r} is not WordPiece")
            return dict(model["vocab"])
        with open(path, encoding="utf-8") as f:
            return {line.rstrip("\n"): i for i, line in enumerate(f)}

    # -- basic tokenisation --------------------------------------------------------------------
    def _basic(self, text: str) -> List[str]:
        if text.isascii():
            # the same rules restricted to ASCII, where they collapse to table lookups: control characters (Cc: 0-31 and 127, minus \t \n
            # \r) vanish, whitespace is " \t\n\r", there is nothing to normalise or to strip accents from, punctuation is the four ASCII
            # ranges of _is_punct (a search query costs 5 us here instead of 45; equality with the general path is fuzzed in
            # tests/test_tokenizers.py)
            text = text.translate(_ASCII_DROP_CONTROLS)
            return _ASCII_BASIC.findall(text.lower() if self.lower else text)
        chars = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            if _is_ws(ch):
                chars.append(" ")
            elif _is_cjk(cp):
                chars.extend((" ", ch, " "))
            else:
                chars.append(ch)
        # NFC normalisation of the raw text mirrors the slow BertTokenizer (prevents treating the same char
        # with different unicode codepoints as different characters)
        words = unicodedata.normalize("NFC", "".join(chars)).split()
        out: List[str] = []
        for w in words:
            if self.lower:
                w = w.lower()
                w = "".join(c for c in unicodedata.normalize("NFD", w) if unicodedata.category(c) != "Mn")
            cur = ""
            for ch in w:
                if _is_punct(ch):
                    if cur:
                        out.append(cur)
                        cur = ""
                    out.append(ch)
                else:
                    cur += ch
            if cur:
                out.append(cur)
        return out

    def _wordpiece(self, word: str) -> List[int]:
        hit = self._cache.get(word)
        if hit is not None:
            return hit
        if len(word) > self.max_chars:
            ids = [self.unk_id]
        else:
            ids, start, n = [], 0, len(word)
            while start < n:
                end, found = n, None
                while start < end:
                    piece = word[start:end] if start == 0 else "##" + word[start:end]
                    if piece in self.vocab:
                        found = self.vocab[piece]
                        break
                    end -= 1
                if found is None:
                    ids = [self.unk_id]
                    break
                ids.append(found)
                start = end
        self._cache[word] = ids
        return ids

    def encode(self, text: str, max_length: Optional[int] = None) -> List[int]:
        ids: List[int] = []
        for part in self._special_split.split(text):
            if not part:
                continue
            if part in self._specials:
                ids.append(self.vocab[part])
                continue
            for w in self._basic(part):
                ids.extend(self._wordpiece(w))
        if max_length is not None and len(ids) > max_length - 2:
            ids = ids[:max(max_length - 2, 0)]
        return [self.cls_id] + ids + [self.sep_id]

    def __call__(self, texts: Union[str, Sequence[str]], max_length: Optional[int] = None) -> Dict[str, np.ndarray]:
        """padding=True (to the longest in the batch), truncation=True -> {'input_ids', 'attention_mask'} int64 [n, S]."""
        if isinstance(texts, str):
            texts = [texts]
        enc = [self.encode(t, max_length) for t in texts]
        S = max((len(e) for e in enc), default=0)
        ids = np.full((len(enc), S), self.pad_id, dtype=np.int64)
        mask = np.zeros((len(enc), S), dtype=np.int64)
        for i, e in enumerate(enc):
            ids[i, :len(e)] = e
            mask[i, :len(e)] = 1
        return {"input_ids": ids, "attention_mask": mask}


# ---------------------------------------------------------------------------------------------------
# RoBERTa byte-level BPE (the text tower of open_clip/roberta-ViT-B-32: HFTokenizer("roberta-base"))
# ---------------------------------------------------------------------------------------------------
class RobertaBpeTokenizer:
    """GPT-2 / RoBERTa byte-level BPE (transformers RobertaTokenizer, third-party and un-vendored; restated from its published
    algorithm and pinned against transformers.RobertaTokenizer in tests/test_tokenizers.py): the text is split by GPT-2's pre-token
    pattern (a leading space belongs to the word that follows), every pre-token's UTF-8 bytes are mapped to printable code points,
    adjacent pairs are merged lowest rank first, pieces are looked up in vocab.json; a row is <s> pieces </s>, truncated to max_length,
    padded with <pad>.  Case is kept.  Special-token spellings inside a text map to their ids (<mask> absorbs the space before it)."""
    PAT = r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""

    def __init__(self, directory: str):
        with open(os.path.join(directory, "vocab.json"), encoding="utf-8") as f:
            self.encoder: Dict[str, int] = json.load(f)
        with open(os.path.join(directory, "merges.txt"), encoding="utf-8") as f:
            lines = [l for l in f.read().split("\n") if l and not l.startswith("#version")]
        self.rank = {tuple(l.split()): i for i, l in enumerate(lines)}
        for t in ("<s>", "<pad>", "</s>", "<unk>"):
            if t not in self.encoder:
                raise ValueError(f"vocab.json has no {t} token")
        self.cls_id, self.pad_id, self.sep_id, self.unk_id = (self.encoder[t] for t in ("<s>", "<pad>", "</s>", "<unk>"))
        self._pat = re.compile(self.PAT)
        specials = [t for t in ("<s>", "<pad>", "</s>", "<unk>", "<mask>") if t in self.encoder]
        self._special_split = re.compile("(" + "|".join((r" ?" if t == "<mask>" else "") + re.escape(t) for t in specials) + ")")
        self._specials = set(specials)
        self._cache: Dict[str, List[int]] = {}

    @property
    def vocab_size(self) -> int:
        return len(self.encoder)

    def _bpe_ids(self, token: str) -> List[int]:
        hit = self._cache.get(token)
        if hit is not None:
            return hit
        b2u = _byte_to_unicode()
        word = [b2u[b] for b in token.encode("utf-8")]
        while len(word) > 1:
            best, best_rank = None, None
            for pair in zip(word[:-1], word[1:]):
                r = self.rank.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            merged, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and word[i] == best[0] and word[i + 1] == best[1]:
                    merged.append(word[i] + word[i + 1])
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        ids = [self.encoder.get(w, self.unk_id) for w in word]
        if len(self._cache) < 1 << 18:
            self._cache[token] = ids
        return ids

    def encode(self, text: str, max_length: Optional[int] = None) -> List[int]:
        ids: List[int] = []
        for part in self._special_split.split(text):
            if not part:
                continue
            if part.lstrip(" ") in self._specials and (part in self._specials or part.lstrip(" ") == "<mask>"):
                ids.append(self.encoder[part.lstrip(" ")])
                continue
            for tok in self._pat.findall(part):
                ids.extend(self._bpe_ids(tok))
        if max_length is not None and len(ids) > max_length - 2:
            ids = ids[:max(max_length - 2, 0)]
        return [self.cls_id] + ids + [self.sep_id]

    def __call__(self, texts: Union[str, Sequence[str]], max_length: Optional[int] = None) -> Dict[str, np.ndarray]:
        if isinstance(texts, str):
            texts = [texts]
        enc = [self.encode(t, max_length) for t in texts]
        S = max((len(e) for e in enc), default=0)
        ids = np.full((len(enc), S), self.pad_id, dtype=np.int64)
        mask = np.zeros((len(enc), S), dtype=np.int64)
        for i, e in enumerate(enc):
            ids[i, :len(e)] = e
            mask[i, :len(e)] = 1
        return {"input_ids": ids, "attention_mask": mask}


# ---------------------------------------------------------------------------------------------------
# XLM-RoBERTa SentencePiece (multilingual-e5 family)
# ---------------------------------------------------------------------------------------------------
class XlmRobertaTokenizer:
    """transformers.XLMRobertaTokenizer over the checkpoint's `sentencepiece.bpe.model` (a unigram SentencePiece model; the
    `sentencepiece` wheel is the third-party dependency the reference reaches through AutoTokenizer, hugging_face_model.py:125-130).
    fairseq id layout: <s> 0, <pad> 1, </s> 2, <unk> 3, then every SentencePiece id shifted by one (SentencePiece's own id 0 =
    <unk> maps to 3); a sequence is <s> pieces... </s>, truncated to max_length, padded to the longest with <pad>."""
    FAIRSEQ_OFFSET = 1

    def __init__(self, model_file: str):
        import sentencepiece as spm
        if os.path.isdir(model_file):
            model_file = os.path.join(model_file, "sentencepiece.bpe.model")
        self.sp = spm.SentencePieceProcessor(model_file=model_file)
        self.cls_id, self.pad_id, self.sep_id, self.unk_id = 0, 1, 2, 3
        self._fairseq = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3}
        self.vocab_size = len(self.sp) + self.FAIRSEQ_OFFSET + 1  # + <mask>

    def _piece_id(self, piece: str) -> int:
        if piece in self._fairseq:
            return self._fairseq[piece]
        i = self.sp.PieceToId(piece)
        return i + self.FAIRSEQ_OFFSET if i else self.unk_id

    def encode(self, text: str, max_length: Optional[int] = None) -> List[int]:
        ids = [self._piece_id(p) for p in self.sp.encode(text, out_type=str)]
        if max_length is not None and len(ids) > max_length - 2:
            ids = ids[:max(max_length - 2, 0)]
        return [self.cls_id] + ids + [self.sep_id]

    def __call__(self, texts: Union[str, Sequence[str]], max_length: Optional[int] = None) -> Dict[str, np.ndarray]:
        if isinstance(texts, str):
            texts = [texts]
        enc = [self.encode(t, max_length) for t in texts]
        S = max((len(e) for e in enc), default=0)
        ids = np.full((len(enc), S), self.pad_id, dtype=np.int64)
        mask = np.zeros((len(enc), S), dtype=np.int64)
        for i, e in enumerate(enc):
            ids[i, :len(e)] = e
            mask[i, :len(e)] = 1
        return {"input_ids": ids, "attention_mask": mask}


# ---------------------------------------------------------------------------------------------------
# stand-in for random-init models (no vocabulary exists for them)
# ---------------------------------------------------------------------------------------------------
def canonicalize_text(text: str) -> str:
    """open_clip's `canonicalize_text` (tokenizer.py; the `clean: canonicalize` of the SigLIP model configs, after big_vision):
    underscores to spaces, every `string.punctuation` character removed, lower-cased, whitespace runs collapsed, stripped."""
    text = text.replace("_", " ").translate(_PUNCT_TABLE).lower()
    return " ".join(text.split())


_PUNCT_TABLE = str.maketrans("", "", __import__("string").punctuation)


class SiglipTokenizer:
    """open_clip HFTokenizer over the SigLIP checkpoints' T5-style SentencePiece vocabulary (32 000 pieces; `timm/ViT-B-16-SigLIP`
    and the Marqo fashion / e-commerce repos ship `tokenizer.json` + `spiece.model`): canonicalize -> pieces + </s> (id 1),
    truncated to and padded (with </s>, the pad token of these tokenizers) to context_length = 64 -> int64 [n, 64].
    `tokenizer.json` is read with the `tokenizers` wheel (what AutoTokenizer's fast path runs); a bare `spiece.model` with
    `sentencepiece`."""

    def __init__(self, path: str, context_length: int = 64):
        self.context_length = context_length
        self.eos_id = self.pad_id = 1
        self._fast = self._sp = None
        d = path if os.path.isdir(path) else os.path.dirname(path)
        tj = path if path.endswith(".json") else os.path.join(d, "tokenizer.json")
        sm = path if path.endswith(".model") else os.path.join(d, "spiece.model")
        if os.path.isfile(tj) and not path.endswith(".model"):
            from tokenizers import Tokenizer
            self._fast = Tokenizer.from_file(tj)
            self._fast.enable_truncation(max_length=context_length)
            self._fast.no_padding()
            self.vocab_size = self._fast.get_vocab_size()
            if os.path.isfile(sm):  # the SentencePiece model beside it feeds the device tokeniser's tables (engine/gpu_tokenizers.py)
                import sentencepiece as spm
                self._sp = spm.SentencePieceProcessor(model_file=sm)
        elif os.path.isfile(sm):
            import sentencepiece as spm
            self._sp = spm.SentencePieceProcessor(model_file=sm)
            self.vocab_size = len(self._sp)
        else:
            raise FileNotFoundError(f"no tokenizer.json / spiece.model under {d}")

    def encode(self, text: str) -> List[int]:
        text = canonicalize_text(text)
        if self._fast is not None:
            return list(self._fast.encode(text).ids)
        ids = list(self._sp.encode(text))[: self.context_length - 1]
        return ids + [self.eos_id]

    def __call__(self, texts: Union[str, Sequence[str]]) -> np.ndarray:
        if isinstance(texts, str):
            texts = [texts]
        out = np.full((len(texts), self.context_length), self.pad_id, dtype=np.int64)
        for i, t in enumerate(texts):
            ids = self.encode(t)
            out[i, :len(ids)] = ids
        return out


class SyntheticTokenizer:
    """Deterministic whitespace-word hash -> id map.  kind='clip': [n, ctx] SOT ... EOT zero-padded with EOT the
    largest id (so argmax pooling finds it); kind='siglip': [n, ctx] words ... </s> padded with </s> (id 1);
    kind='bert': CLS ... SEP, padded to the longest, with a mask."""

    def __init__(self, kind: str, vocab_size: int, context_length: int = 77):
        assert kind in ("clip", "bert", "siglip", "xlmr")
        self.kind, self.vocab_size, self.context_length = kind, vocab_size, context_length
        self._ids: Dict[str, int] = {}
        if kind == "xlmr":   # XLM-RoBERTa framing (<s> 0 ... </s> 2, <pad> 1), otherwise the 'bert' behaviour
            self.kind = "bert"
            self.cls_id, self.sep_id, self.pad_id, self.lo, self.hi = 0, 2, 1, 4, vocab_size
            return
        if kind == "clip":
            self.sot_id, self.eot_id, self.lo, self.hi = vocab_size - 2, vocab_size - 1, 1, vocab_size - 2
        elif kind == "siglip":
            self.eos_id, self.lo, self.hi = 1, 2, vocab_size
        else:
            self.cls_id, self.sep_id, self.pad_id, self.lo, self.hi = 101, 102, 0, 1000, vocab_size

    def _word_id(self, w: str) -> int:
        hit = self._ids.get(w)
        if hit is None:
            h = int.from_bytes(hashlib.blake2b(w.encode("utf-8"), digest_size=8).digest(), "little")
            hit = self.lo + h % (self.hi - self.lo)
            if len(self._ids) < 1 << 16:
                self._ids[w] = hit
        return hit

    def encode_words(self, text: str) -> List[int]:
        return [self._word_id(w) for w in text.lower().split()]

    def __call__(self, texts, max_length: Optional[int] = None):
        if isinstance(texts, str):
            texts = [texts]
        if self.kind == "siglip":
            L = self.context_length
            out = np.full((len(texts), L), self.eos_id, dtype=np.int64)
            for i, t in enumerate(texts):
                ids = self.encode_words(canonicalize_text(t))[:L - 1]
                out[i, :len(ids)] = ids
            return out
        if self.kind == "clip":
            L = self.context_length
            out = np.zeros((len(texts), L), dtype=np.int64)
            for i, t in enumerate(texts):
                ids = [self.sot_id] + self.encode_words(t) + [self.eot_id]
                if len(ids) > L:
                    ids = ids[:L]
                    ids[-1] = self.eot_id
                out[i, :len(ids)] = ids
            return out
        enc = []
        for t in texts:
            ids = self.encode_words(t)
            if max_length is not None:
                ids = ids[:max(max_length - 2, 0)]
            enc.append([self.cls_id] + ids + [self.sep_id])
        S = max((len(e) for e in enc), default=0)
        ids = np.full((len(enc), S), self.pad_id, dtype=np.int64)
        mask = np.zeros((len(enc), S), dtype=np.int64)
        for i, e in enumerate(enc):
            ids[i, :len(e)] = e
            mask[i, :len(e)] = 1
        return {"input_ids": ids, "attention_mask": mask}
