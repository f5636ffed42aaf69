"""Independent sentence / word segmenters for the reference run (TEST INFRASTRUCTURE ONLY — nothing under marqo_amd/ imports this).

The reference splits with nltk punkt (`sent_tokenize` / `word_tokenize`, src/marqo/s2_inference/processing/text.py:109-160); nltk and its punkt
model are not installable here, so the run of the reference's own `split_text` (tests/golden/make_ref_golden.py, tests/ref_suite_runner.py)
needs a segmenter injected through oracle/ref_shim.py.  Round 2 injected the PRODUCT's own regex splitters, which made the segmentation
half of that parity statement circular.  These are a second, independently written implementation of the same published boundary rules —
character scanners, no regular expressions, no code shared with marqo_amd/s2_inference/processing/text.py — so that agreement between the
reference-run fixtures and the product is agreement between two implementations, not an identity.

Rules (the punkt behaviour both sides document as their contract; unusual abbreviations are the stated deviation of both):
  sentence boundary = one of . ! ?  [+ any closing quotes / brackets]  + whitespace, when the next sentence starts with an optional opening
                      quote / bracket followed by an upper-case letter or a digit;
  word tokens       = maximal runs of word characters (letters, digits, underscore) with at most one internal apostrophe group (don't),
                      every other non-space character is a token of its own.
"""
from typing import List

_CLOSERS = "\"')]"
_OPENERS = "\"'(["


def _is_word_char(ch: str) -> bool:
    return ch == "_" or ch.isalnum()


def sentences(text: str) -> List[str]:
    out, start, i, n = [], 0, 0, len(text)
    while i < n:
        ch = text[i]
        if ch in ".!?":
            j = i + 1
            while j < n and text[j] in _CLOSERS:          # closing quotes / brackets stay with the sentence
                j += 1
            k = j
            while k < n and text[k].isspace():
                k += 1
            if k > j and k < n:                            # some whitespace, and something follows
                m = k + 1 if text[k] in _OPENERS and k + 1 < n else k
                nxt = text[m]
                if (nxt.isascii() and (nxt.isupper() or nxt.isdigit())):
                    piece = text[start:j].strip()
                    if piece:
                        out.append(piece)
                    start = k
                    i = k
                    continue
        i += 1
    tail = text[start:].strip()
    if tail:
        out.append(tail)
    return out


def words(text: str) -> List[str]:
    out, i, n = [], 0, len(text)
    while i < n:
        ch = text[i]
        if ch.isspace():
            i += 1
        elif _is_word_char(ch):
            j = i + 1
            while j < n and _is_word_char(text[j]):
                j += 1
            if j + 1 < n and text[j] == "'" and _is_word_char(text[j + 1]):   # one apostrophe group: don't, o'clock
                j += 2
                while j < n and _is_word_char(text[j]):
                    j += 1
            out.append(text[i:j])
            i = j
        else:
            out.append(ch)
            i += 1
    return out
