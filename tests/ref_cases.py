"""Deterministic inputs shared by the reference-run fixture generator (tests/golden/make_ref_golden.py, which executes the
REFERENCE'S OWN functions under oracle/ref_shim.py) and the parity tests that replay the same inputs through the product
(tests/test_ref_parity.py, tests/test_ref_parity_gpu.py).  Nothing here depends on the reference."""
import numpy as np
import torch
from PIL import Image

from oracle import towers as O

IMAGE_SIZES = [(64, 64), (100, 80), (70, 200), (240, 240), (333, 500), (31, 17)]   # (h, w)

PATCH_METHODS = ["simple", "overlap", "simple?hn=2&wn=4", "overlap?hn=2&wn=2", "simple?hn=1&wn=1", "simple?hn=5&wn=7"]

SPLIT_TEXT = ("Marqo is a tensor search engine. It embeds text and images! Does it scale? Yes. "
              "Each document is split into chunks; every chunk is vectorised. The end.\n\nSecond passage starts here. It is short.\n\nThird.")

SPLIT_CASES = [  # (split_by, split_length, split_overlap)
    ("character", 10, 3), ("character", 128, 16), ("character", 7, 0), ("word", 5, 2), ("word", 3, 0), ("word", 50, 10),
    ("sentence", 2, 1), ("sentence", 2, 0), ("sentence", 3, 1), ("sentence", 1, 0), ("sentence", 20, 5),
    ("passage", 1, 0), ("passage", 2, 1), ("passage", 5, 2),
]
SPLIT_EDGE_TEXTS = ["", " ", "a", "One sentence only.", "No terminal punctuation", "   \n  ", "A. B. C. D. E. F. G.",
                    # closing quotes / brackets belong to the sentence they close (round 3: the product used to drop them)
                    'He said "Go." Then he left. (Really.) [Ok.] \'Fine.\' Next one, "quoted start." 3 more.',
                    "lower case after a stop. stays one sentence? yes! It's 5 o'clock. Don't split o'clock."]

WRAPPER_TEXTS = ["a photo of a cat", "The Quick  Brown fox, jumps over the lazy dog!", "query: how much protein should a female eat",
                 "marqo is a tensor search engine", "it's built for images and text", "dog"]

MODEL_SIZE_CASES = [  # (model_name, model_properties)
    ("open_clip/ViT-B-32/laion2b_s34b_b79k", {"name": "open_clip/ViT-B-32/laion2b_s34b_b79k", "dimensions": 512, "type": "open_clip"}),
    ("open_clip/ViT-L-14/laion2b_s32b_b82k", {"name": "open_clip/ViT-L-14/laion2b_s32b_b82k", "dimensions": 768, "type": "open_clip"}),
    ("open_clip/ViT-H-14/laion2b_s32b_b79k", {"name": "x", "dimensions": 1024, "type": "open_clip"}),
    ("open_clip/ViT-g-14/laion2b_s12b_b42k", {"name": "x", "dimensions": 1024, "type": "open_clip"}),
    ("open_clip/ViT-bigG-14/laion2b_s39b_b160k", {"name": "x", "dimensions": 1280, "type": "open_clip"}),
    ("ViT-B/32", {"name": "ViT-B/32", "dimensions": 512, "type": "clip"}),
    ("hf/e5-base-v2", {"name": "intfloat/e5-base-v2", "dimensions": 768, "tokens": 512, "type": "hf"}),
    ("sentence-transformers/all-MiniLM-L6-v1", {"name": "all-MiniLM-L6-v1", "dimensions": 384, "type": "sbert"}),
    ("random/small", {"name": "random/small", "dimensions": 32, "tokens": 128, "type": "random"}),
    ("my-model", {"name": "my-model", "dimensions": 10, "type": "open_clip", "model_size": 7.5}),
    ("unknown-type", {"name": "q", "dimensions": 10, "type": "something_else"}),
    ("no-type", {"name": "q", "dimensions": 10}),
]

CACHE_KEY_CASES = [
    ("ViT-B/32", "cpu", None),
    ("hf/e5-base-v2", "cuda:0", {"name": "intfloat/e5-base-v2", "dimensions": 768, "tokens": 512, "type": "hf"}),
    ("my-model", "cuda", {"name": "hf-hub:acme/tiny", "dimensions": 64, "type": "open_clip"}),
    ("random/small", "cuda:1", {"dimensions": 32, "type": "random"}),
]

IS_IMAGE_CASES = [  # (label, spec) — spec is rebuilt into the actual object by is_image_input()
    ("jpg_name", ("str", "cat.jpg")), ("png_upper", ("str", "DIR/Cat.PNG")), ("jpeg_url", ("str", "https://a.b/c.jpeg?x=1")),
    ("bmp", ("str", "x.bmp")), ("gif_name", ("str", "x.gif")), ("url_no_ext", ("str", "https://marqo.ai/image")),
    ("http_url", ("str", "http://example.com/a/b")), ("plain_text", ("str", "a photo of a cat")), ("empty_str", ("str", "")),
    ("dotted_text", ("str", "hello.world this is text")), ("ftp_like", ("str", "not a url://x")),
    ("list_first_img", ("list", ["a.png", "some text"])), ("list_first_text", ("list", ["some text", "a.png"])),
    ("empty_list", ("list", [])), ("pil", ("pil", None)), ("ndarray", ("ndarray", None)), ("tensor", ("tensor", None)),
    ("list_pil", ("list_pil", None)), ("int", ("int", 3)), ("none", ("none", None)), ("list_int", ("list", [1, 2])),
]

RANDOM_CASES = [("random/small", 32), ("random/medium", 128), ("random", 384)]
RANDOM_INPUTS = ["hello", ["hello"], ["hello", "world"], ["a", "b", "c", "hello"], "", ["", ""]]

CONVERT_CASES = ["nd_2d", "nd_1d", "tensor_2d", "tensor_1d", "list_2d", "nd_f16", "tensor_f64"]

TINY_BERT = O.BertConfig(vocab=0, max_pos=64, width=128, layers=2, heads=2, mlp_dim=512)          # vocab filled in from the vocabulary
TINY_VIT = O.VitConfig(image_size=64, patch_size=16, width=128, layers=2, heads=2, mlp_dim=256, out_dim=64)
TINY_TEXT = O.ClipTextConfig(vocab=0, ctx=77, width=128, layers=2, heads=2, mlp_dim=256, out_dim=64)


def images():
    rng = np.random.default_rng(20240923)
    return [Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for h, w in IMAGE_SIZES]


def is_image_input(spec):
    kind, val = spec
    if kind in ("str", "list", "int"):
        return val
    if kind == "none":
        return None
    rng = np.random.default_rng(1)
    arr = rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)
    if kind == "pil":
        return Image.fromarray(arr)
    if kind == "ndarray":
        return arr
    if kind == "tensor":
        return torch.zeros(3, 8, 8)
    if kind == "list_pil":
        return [Image.fromarray(arr), "text"]
    raise ValueError(kind)


def convert_input(name):
    rng = np.random.default_rng(7)
    a = rng.standard_normal((3, 5)).astype(np.float32)
    return {"nd_2d": a, "nd_1d": a[0], "tensor_2d": torch.from_numpy(a), "tensor_1d": torch.from_numpy(a[1]),
            "list_2d": a.tolist(), "nd_f16": a.astype(np.float16), "tensor_f64": torch.from_numpy(a.astype(np.float64))}[name]


def bert_vocab():
    from tests.test_tokenizers import _bert_vocab
    return _bert_vocab()


def clip_merges():
    from tests.test_tokenizers import CORPUS, _train_bpe
    return _train_bpe(CORPUS, 120)


def tiny_bert_cfg():
    cfg = O.BertConfig(**{**TINY_BERT.__dict__, "vocab": len(bert_vocab())})
    return cfg


def tiny_text_cfg():
    return O.ClipTextConfig(**{**TINY_TEXT.__dict__, "vocab": 512 + len(clip_merges()) + 2})
