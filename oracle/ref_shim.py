"""Import shim that runs the REFERENCE'S OWN Python modules in this container.  TEST INFRASTRUCTURE ONLY.

The reference (`/root/reference/src/marqo`) is a Python package whose import chain pulls in wheels that are not installed here
(torchvision, open_clip, clip, sentence_transformers, cv2, onnxruntime, pycurl, validators, nltk, more_itertools, semver, boto3,
fastapi-on-pydantic-v1 ...) and that is written against pydantic v1.  None of those wheels do arithmetic on the hot path's
*wrapper* code — the functions SURVEY.md §8(a) cites (`vectorise` plumbing, `_is_image`, `chunk_image` / `generate_boxes`,
`split_text`, `Random.encode`, `HuggingFaceModel.encode` + pooling, `OPEN_CLIP.encode_image/encode_text`, the registry dict) are
plain Python / numpy / torch / PIL.  `install()` therefore

  * aliases `pydantic` to the `pydantic.v1` compatibility namespace that pydantic 2 ships,
  * registers a meta-path finder that satisfies imports of the missing wheels with inert stub modules (every attribute is an
    inert class; nothing in them computes), and of the reference's out-of-scope `languagebind` package (video / audio),
  * gives FUNCTIONAL stand-ins only for the three third-party callables the cited functions actually execute:
      - `more_itertools.windowed`  (published semantics: sliding window with `fillvalue=None` padding of the last window),
      - `validators.url`           (truthy for `scheme://host...` URLs, falsy otherwise),
      - `nltk.tokenize.sent_tokenize / word_tokenize` (punkt data cannot be downloaded; the caller injects the segmenter, so the
        reference's windowing / re-joining code runs on the same segmentation as the product),
  * puts `/root/reference/src` on sys.path with bytecode writing OFF (the reference tree is read-only by contract).

Because it rebinds `pydantic` process-wide it must only ever be installed in a dedicated subprocess
(`tests/golden/make_ref_golden.py` is that process; `tests/test_ref_parity.py` re-runs it when `/root/reference` exists and
compares with the committed fixtures).  Nothing under `marqo_amd/` imports this module.
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import os
import re
import sys
import types
from typing import Callable, List, Optional

REFERENCE_SRC = os.environ.get("MARQO_REFERENCE_SRC", "/root/reference/src")

# top-level packages the reference imports that are absent from this image (or, for fastapi / starlette, present but bound to
# pydantic 2): all of them belong to the reference's control plane, model download or out-of-scope model families
STUB_ROOTS = {
    "cv2", "onnxruntime", "onnx", "optimum", "torchvision", "open_clip", "clip", "sentence_transformers", "pycurl", "validators",
    "magic", "nltk", "more_itertools", "multilingual_clip", "semver", "kazoo", "cachetools", "readerwriterlock", "ftfy", "timm",
    "jsonschema", "redis", "pympler", "orjson", "uvicorn", "fastapi", "starlette", "boto3", "botocore", "ffmpeg", "decord",
    "pytorchvideo", "torchaudio",
}
STUB_PREFIXES = ("marqo.s2_inference.languagebind",)


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "marqo"))


class _Inert:
    """instance returned by calling / indexing any stub attribute"""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __getattr__(self, n):
        if n.startswith("__") and n.endswith("__"):
            raise AttributeError(n)
        return _Inert()

    def __getitem__(self, k):
        return _Inert()

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _InertMeta(type):
    def __getattr__(cls, n):
        if n.startswith("__") and n.endswith("__"):
            raise AttributeError(n)
        return _Inert()


class _InertBase(metaclass=_InertMeta):
    """stub attributes are CLASSES (the reference subclasses / isinstance-checks a few of them, e.g. `Compose`)"""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __getattr__(self, n):
        if n.startswith("__") and n.endswith("__"):
            raise AttributeError(n)
        return _Inert()


class _StubModule(types.ModuleType):
    __path__: List[str] = []

    def __getattr__(self, n):
        if n.startswith("__") and n.endswith("__"):
            raise AttributeError(n)
        v = type(n, (_InertBase,), {})
        setattr(self, n, v)
        return v


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in STUB_ROOTS or name.startswith(STUB_PREFIXES):
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        hook = _FUNCTIONAL.get(module.__name__)
        if hook is not None:
            hook(module)


def _windowed(seq, n, fillvalue=None, step=1):
    """more_itertools.windowed as published (more-itertools 8-10): windows of width n advancing by `step`; if the iterable is
    shorter than n one padded window is produced; a trailing partial window is produced, padded with `fillvalue`, when the
    remaining items do not line up with the step."""
    if n < 0:
        raise ValueError("n must be >= 0")
    if n == 0:
        yield ()
        return
    if step < 1:
        raise ValueError("step must be >= 1")
    from collections import deque
    window = deque(maxlen=n)
    i = n
    for _ in map(window.append, seq):
        i -= 1
        if not i:
            i = step
            yield tuple(window)
    size = len(window)
    if size == 0:
        return
    elif size < n:
        yield tuple(window) + ((fillvalue,) * (n - size))
    elif 0 < i < min(step, n):
        window += (fillvalue,) * i
        yield tuple(window)


_URL = re.compile(r"^(?:https?|ftp)://[^\s/?#]+\.[^\s/?#]+(?:[/?#]\S*)?$|^(?:https?|ftp)://(?:localhost|\d{1,3}(?:\.\d{1,3}){3})(?::\d+)?(?:[/?#]\S*)?$",
                  re.IGNORECASE)

_segmenters = {"sent": None, "word": None}


def _fn_more_itertools(m):
    m.windowed = _windowed


def _fn_validators(m):
    m.url = lambda value, *a, **k: bool(isinstance(value, str) and _URL.match(value))


def _fn_nltk(m):
    class _Data:
        @staticmethod
        def find(*a, **k):
            return "stub"
    m.data = _Data
    m.download = lambda *a, **k: True


def _fn_nltk_tokenize(m):
    def sent_tokenize(text, language="english"):
        if _segmenters["sent"] is None:
            raise LookupError("ref_shim: no sentence segmenter injected (punkt data is not downloadable here)")
        return list(_segmenters["sent"](text))

    def word_tokenize(text, language="english", preserve_line=False):
        if _segmenters["word"] is None:
            raise LookupError("ref_shim: no word segmenter injected (punkt data is not downloadable here)")
        return list(_segmenters["word"](text))
    m.sent_tokenize, m.word_tokenize = sent_tokenize, word_tokenize


_FUNCTIONAL = {"more_itertools": _fn_more_itertools, "validators": _fn_validators, "nltk": _fn_nltk, "nltk.tokenize": _fn_nltk_tokenize}

_installed = False


def install(sent_tokenize: Optional[Callable[[str], List[str]]] = None,
            word_tokenize: Optional[Callable[[str], List[str]]] = None) -> None:
    """Make `import marqo...` resolve to the reference's own source tree.  Call once, first thing, in a dedicated process."""
    global _installed
    _segmenters["sent"], _segmenters["word"] = sent_tokenize, word_tokenize
    if _installed:
        return
    if not available():
        raise FileNotFoundError(f"reference source tree not found at {REFERENCE_SRC}")
    sys.dont_write_bytecode = True  # never write __pycache__ into the (read-only) reference tree
    # transformers probes optional packages with importlib.util.find_spec; let it look at the REAL environment before the stub
    # finder exists, and materialise the lazily imported classes the reference's `from transformers import ...` lines need
    import transformers
    for name in ("AutoModel", "AutoTokenizer", "AutoConfig", "CLIPModel", "CLIPProcessor"):
        getattr(transformers, name, None)
    import pydantic.v1 as pv1
    sys.modules["pydantic"] = pv1
    for sub in ("error_wrappers", "fields", "main", "typing", "class_validators", "generics", "errors", "types", "validators"):
        try:
            sys.modules["pydantic." + sub] = importlib.import_module("pydantic.v1." + sub)
        except ImportError:
            pass
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, REFERENCE_SRC)
    import warnings
    warnings.filterwarnings("ignore")
    _installed = True
