"""Text chunking before vectorise (host strings; reference: src/marqo/s2_inference/processing/text.py:9-177).

`split_text` windows a text by character / word / sentence / passage with (split_length, split_overlap) and
`prefix_text_chunks` prepends the model's chunk prefix (e.g. "passage: " for e5, model_registry.py:776-777).
The reference tokenises words / sentences with nltk punkt (an un-vendored model that needs a download); when nltk
with its punkt data is importable it is used, otherwise a rule-based splitter stands in (documented deviation: only
the sentence boundaries of unusual abbreviations can differ).
"""
from __future__ import annotations

import re
from typing import Callable, List, Optional

_SENT_END = re.compile(r"""([.!?]["')\]]*)(\s+)(?=["'(\[]?[A-Z0-9])""")
_WORD = re.compile(r"\w+(?:'\w+)?|[^\w\s]")


def _nltk_tokenizers(language: str):
    try:
        import nltk
        from nltk.tokenize import sent_tokenize, word_tokenize
        nltk.data.find("tokenizers/punkt")
        return (lambda t: sent_tokenize(t, language=language)), (lambda t: word_tokenize(t, language=language))
    except Exception:
        return None


def _sentences(text: str) -> List[str]:
    """sentence boundary = . ! ? [+ closing quotes / brackets, which STAY with their sentence] + whitespace before an (optionally quoted /
    bracketed) upper-case letter or digit.  (Round 3: the closers used to be swallowed with the whitespace — `He said "Go." Then left.` lost
    its closing quote; found by the independently written segmenter of oracle/segment.py, which the reference run now uses.)"""
    out, start = [], 0
    for m in _SENT_END.finditer(text):
        piece = text[start:m.end(1)].strip()
        if piece:
            out.append(piece)
        start = m.end(2)
    tail = text[start:].strip()
    if tail:
        out.append(tail)
    return out


def _splitting_functions(split_by: str, language: str = "english") -> Callable[[str], List[str]]:
    if not isinstance(split_by, str):
        raise TypeError(f"expected str received {type(split_by)}")
    nl = _nltk_tokenizers(language)
    mapping = {
        "character": list,
        "word": nl[1] if nl else _WORD.findall,
        "sentence": nl[0] if nl else _sentences,
        "passage": lambda x: x.split("\n\n"),
    }
    if split_by in mapping:
        return mapping[split_by]
    raise KeyError(f"unexpected split_by type of {split_by}")


def _windowed(seq: List, n: int, step: int) -> List[List]:
    """more_itertools.windowed(seq, n, step) with its None fill (dropped on re-join)."""
    if n < 0:
        raise ValueError("n must be >= 0")
    if n == 0:
        return [[]]
    if step < 1:
        raise ValueError("step must be >= 1")
    out, i = [], 0
    if not seq:
        return [[None] * n]
    while True:
        w = seq[i:i + n]
        if len(w) < n:
            if i == 0 or len(w) > n - step:
                out.append(w + [None] * (n - len(w)))
            break
        out.append(w)
        if i + n >= len(seq):
            break
        i += step
    return out


def check_make_string_valid(text: str, coerce: bool = True) -> str:
    empty_string = " "
    if text in [[], None, "", empty_string] and coerce:
        return empty_string
    if text.isspace():   # (before the type check, as text.py:97-103 has it: a non-str raises AttributeError here)
        return empty_string
    if not isinstance(text, str):
        raise TypeError(f"text had type {type(text)} but expected str")
    return text


def split_text(text: str, split_by: str = "sentence", split_length: int = 2, split_overlap: int = 1,
               language: str = "english", custom_seperator: Optional[str] = None) -> List[str]:
    if split_length == 0:
        raise ValueError("split length must be > 0")
    text = check_make_string_valid(text, coerce=True)
    if len(text) <= 1:
        return [text]
    if custom_seperator is None:
        seperator = "" if split_by == "character" else " "
    else:
        seperator = custom_seperator
    pieces = _splitting_functions(split_by, language=language)(text)
    segments = _windowed(list(pieces), n=split_length, step=split_length - split_overlap)
    results = []
    for seg in segments:
        txt = seperator.join(t for t in seg if t is not None)
        if len(txt) > 0:
            results.append(txt)
    return results


def prefix_text_chunks(text_splits: List[str], text_chunk_prefix: str) -> List[str]:
    if not text_chunk_prefix:
        return text_splits
    return [text_chunk_prefix + t for t in text_splits]
