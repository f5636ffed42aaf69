"""Parity against THE REFERENCE ITSELF (CPU half).

tests/golden/ref_host.json + ref_wrappers.npz are outputs of the reference's own functions, produced by
tests/golden/make_ref_golden.py importing /root/reference/src under oracle/ref_shim.py.  Here:
  * `test_fixtures_reproduce_from_reference` re-runs that script when /root/reference is present (this container) and requires the
    committed fixtures to be byte-equal in content — the fixtures provably come from the reference, not from us;
  * every other test replays the same inputs (tests/ref_cases.py) through the PRODUCT's host code (marqo_amd.s2_inference.*) and
    through the ORACLE (oracle/towers.py, oracle/preprocess.py) and compares with the reference's answers: exact for strings,
    integers, boxes, exception classes and pixels; 1e-6 for the fp32 wrappers (same towers injected, so only the reference's
    pooling / normalisation / conversion code differs from the oracle's restatement).
The GPU half (HIP kernels vs these fixtures) is tests/test_ref_parity_gpu.py.
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
from PIL import UnidentifiedImageError

from oracle import preprocess as OP
from oracle import ref_shim
from oracle import towers as O
from tests import ref_cases as RC

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host():
    with open(os.path.join(GOLDEN, "ref_host.json"), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="module")
def arrays():
    return dict(np.load(os.path.join(GOLDEN, "ref_wrappers.npz")))


def _sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _outcome(fn):
    """same shape as make_ref_golden._exc, exception classes by NAME (product classes mirror the reference's names)"""
    try:
        return {"ok": fn()}
    except Exception as e:  # noqa: BLE001
        return {"raises": type(e).__name__}


def _same_outcome(ref, got, what=""):
    if "raises" in ref:
        assert got.get("raises") == ref["raises"], f"{what}: reference raises {ref['raises']}, product gave {got}"
    else:
        assert "ok" in got, f"{what}: reference returns {ref['ok']!r}, product raised {got}"
        assert got["ok"] == ref["ok"], f"{what}: {got['ok']!r} != reference {ref['ok']!r}"


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference is not present on this machine")
def test_fixtures_reproduce_from_reference(tmp_path, host, arrays):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    for k in ("MARQO_MAX_VECTORISE_BATCH_SIZE", "MARQO_AMD_MODEL_DIR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_ref_golden.py"), "--out", str(tmp_path)], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    with open(tmp_path / "ref_host.json", encoding="utf-8") as f:
        fresh = json.load(f)
    assert fresh == host, "tests/golden/ref_host.json is stale: re-run tests/golden/make_ref_golden.py"
    fresh_arrays = dict(np.load(tmp_path / "ref_wrappers.npz"))
    assert sorted(fresh_arrays) == sorted(arrays)
    for k, v in arrays.items():
        assert v.shape == fresh_arrays[k].shape and v.dtype == fresh_arrays[k].dtype, k
        if v.dtype.kind == "f":
            assert np.allclose(v, fresh_arrays[k], rtol=0, atol=2e-6), k   # (fp32 towers: thread-count dependent summation order)
        else:
            assert np.array_equal(v, fresh_arrays[k]), k


def test_shim_never_reaches_the_product():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "marqo_amd")):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dirpath, fn), encoding="utf-8").read()
                assert "ref_shim" not in src and "/root/reference" not in src, os.path.join(dirpath, fn)
    assert "ref_shim" not in open(os.path.join(ROOT, "bench.py")).read()


# ---- registry ---------------------------------------------------------------------------------------------------------
def test_registry_entries_match_reference(host):
    from marqo_amd.s2_inference.model_registry import load_model_properties
    ours = load_model_properties()
    ref = host["registry_models"]
    checked = 0
    for name, p in ours["models"].items():
        assert name in ref, f"{name} is registered here but not in the reference registry"
        r = ref[name]
        for field in ("name", "dimensions", "type", "tokens", "text_query_prefix", "text_chunk_prefix", "poolingMethod", "pretrained",
                      "trustRemoteCode", "visual_model", "textual_model"):
            if field in r:
                assert p.get(field) == r[field], f"{name}.{field}: {p.get(field)!r} != reference {r[field]!r}"
                checked += 1
            elif field in ("poolingMethod", "text_query_prefix", "text_chunk_prefix", "tokens", "trustRemoteCode"):
                assert field not in p, f"{name}.{field} set here ({p[field]!r}) but absent in the reference"
    assert checked > 300
    assert set(ours["loaders"]) <= set(host["registry_loader_types"])
    # every reference open_clip entry whose architecture is a plain ViT / SigLIP tower is registered
    from marqo_amd.engine import archs
    missing = []
    for name, r in ref.items():
        if r["type"] == "open_clip" and name.startswith("open_clip/") and name not in ours["models"]:
            try:
                archs.resolve_open_clip(name.split("/")[1], name.split("/")[2])
            except Exception:  # noqa: BLE001 - other tower families (ResNet / ConvNeXt / EVA / CoCa / roberta text)
                continue
            missing.append(name)
    assert not missing, f"ViT-family reference entries not registered: {missing}"
    ref_hf_bert = [n for n, r in ref.items() if r["type"] == "hf"]
    assert sum(1 for n in ref_hf_bert if n in ours["models"]) >= 23


# ---- s2_inference plumbing ----------------------------------------------------------------------------------------------
def test_plumbing_known_answers(host, monkeypatch):
    from marqo_amd.s2_inference import s2_inference as s2
    for k, ref in host["generate_batches"].items():
        n, b = map(int, k.split("/"))
        assert [list(x) for x in s2.generate_batches(list(range(n)), b)] == ref
    assert [s2._create_model_cache_key(n, d, p) for n, d, p in RC.CACHE_KEY_CASES] == host["cache_key"]
    assert [s2.get_model_size(n, p) for n, p in RC.MODEL_SIZE_CASES] == host["model_size"]
    for name in RC.CONVERT_CASES:
        ref = host["convert_vectorized_output"][name]
        got = _outcome(lambda: s2._convert_vectorized_output(RC.convert_input(name)))
        _same_outcome(ref, got, name)
    for k, v in {"ok": [[1.0, 2.0]], "int": [[1, 2]], "flat": [1.0, 2.0], "np_float": [[float(np.float32(1.5))]], "empty": []}.items():
        _same_outcome(host["check_output_type"][k], _outcome(lambda: s2._check_output_type(v)), k)
    for val, ref in host["max_vectorise_batch_size"].items():
        if val == "None":
            monkeypatch.delenv("MARQO_MAX_VECTORISE_BATCH_SIZE", raising=False)
        else:
            monkeypatch.setenv("MARQO_MAX_VECTORISE_BATCH_SIZE", val)
        _same_outcome(ref, _outcome(s2._get_max_vectorise_batch_size), f"MARQO_MAX_VECTORISE_BATCH_SIZE={val}")
    monkeypatch.delenv("MARQO_MAX_VECTORISE_BATCH_SIZE", raising=False)


def test_validate_model_properties_matches_reference(host):
    from marqo_amd.s2_inference import s2_inference as s2
    cases = {
        "unknown_no_props": ("not-a-model", None),
        "custom_hf": ("mine", {"name": "a/b", "dimensions": 8, "type": "hf"}),
        "custom_default_type": ("mine", {"name": "a/b", "dimensions": 8}),
        "missing_dims": ("mine", {"name": "a/b", "type": "hf"}), "missing_name": ("mine", {"dimensions": 8, "type": "sbert"}),
        "open_clip_localpath": ("mine", {"dimensions": 8, "type": "open_clip", "localpath": "/x"}),
        "no_model_ok": ("no_model", {"dimensions": 8, "type": "no_model"}), "no_model_no_dims": ("no_model", {"type": "no_model"}),
        "no_model_wrong_name": ("x", {"dimensions": 8, "type": "no_model"}), "bad_dims": ("mine", {"name": "q", "dimensions": -1, "type": "hf"}),
    }
    for label, (name, props) in cases.items():
        ref = host["validate_model_properties"][label]
        got = _outcome(lambda: s2.validate_model_properties(name, None if props is None else dict(props)))
        _same_outcome(ref, got, label)
    # a registry name resolves to the same properties as in the reference (fields the reference defines)
    got = s2.validate_model_properties("hf/e5-base-v2", None)
    ref = host["validate_model_properties"]["registry_name"]["ok"]
    for k in ("name", "dimensions", "tokens", "type", "text_query_prefix", "text_chunk_prefix"):
        assert got[k] == ref[k]


def test_vectorise_with_the_references_random_models(host, arrays):
    """the whole vectorise() flow (registry -> loader -> 16-item batch loop -> concatenate -> tolist) on the reference's own
    `random/*` plumbing fakes gives the reference's numbers"""
    from marqo_amd.s2_inference import s2_inference as s2
    from marqo_amd.s2_inference.random_utils import Random, sentence_to_hash
    for s, h in host["random"]["sentence_to_hash"].items():
        assert sentence_to_hash(s) == h
    for name, dim in RC.RANDOM_CASES:
        m = Random(name, device="cpu", embedding_dim=dim)
        m.load()
        for i, inp in enumerate(RC.RANDOM_INPUTS):
            assert np.array_equal(np.asarray(m.encode(inp), dtype=np.float64), arrays[f"random:{name}:{i}"]), (name, inp)
    s2.clear_loaded_models()
    cases = {
        "list": dict(model_name="random/small", content=["hello", "world", "marqo"], device="cpu"),
        "str": dict(model_name="random/small", content="hello", device="cpu"),
        "n40": dict(model_name="random/medium", content=[f"doc {i}" for i in range(40)], device="cpu"),
        "unnormalised": dict(model_name="random", content=["a", "b"], device="cpu", normalize_embeddings=False),
        "no_device": dict(model_name="random/small", content=["a"]),
        "empty_list": dict(model_name="random/small", content=[], device="cpu"),
        "unknown_model": dict(model_name="definitely/not-a-model", content=["a"], device="cpu"),
        "bad_props": dict(model_name="m", content=["a"], device="cpu", model_properties={"type": "random"}),
    }
    for label, kw in cases.items():
        ref = host["vectorise_random"][label]
        got = _outcome(lambda: s2.vectorise(**kw))
        if "raises" in ref:
            assert got.get("raises") == ref["raises"], (label, got)
        else:
            out = got["ok"]
            assert len(out) == ref["ok"]["n"] and len(out[0]) == ref["ok"]["d"] and type(out[0][0]).__name__ == ref["ok"]["elem_type"]
            assert np.array_equal(np.asarray(out, dtype=np.float64), arrays[f"vectorise:{label}"]), label
    assert sorted(s2.get_available_models().keys()) == host["available_models_after"]
    s2.clear_loaded_models()


# ---- image typing / chunking ----------------------------------------------------------------------------------------------
def test_is_image_matches_reference(host):
    from marqo_amd.s2_inference.image_input import _is_image, format_and_load_CLIP_image
    from PIL import Image
    for label, spec in RC.IS_IMAGE_CASES:
        _same_outcome(host["is_image"][label], _outcome(lambda: bool(_is_image(RC.is_image_input(spec)))), label)
    for label, spec in [("pil", ("pil", None)), ("ndarray", ("ndarray", None)), ("tensor", ("tensor", None)), ("int", ("int", 3))]:
        ref = host["format_and_load_CLIP_image"][label]
        got = _outcome(lambda: format_and_load_CLIP_image(RC.is_image_input(spec), {}))
        if "ok" in got:
            got = {"ok": type(got["ok"]).__name__ if not isinstance(got["ok"], Image.Image) else "PIL:" + got["ok"].mode}
        _same_outcome(ref, got, label)


def test_box_math_matches_reference(host):
    from marqo_amd.s2_inference.processing import image as pi
    for k, ref in host["generate_boxes"].items():
        size, grid, ov = k.split("/")
        w, h = map(int, size.split("x"))
        hn, wn = map(int, grid.split("x"))
        assert [list(b) for b in pi.generate_boxes((w, h), hn, wn, overlap=bool(int(ov)))] == ref, k
        assert [list(b) for b in OP.generate_boxes((w, h), hn, wn, overlap=bool(int(ov)))] == ref, k
    cases = [((0, 0, 80, 80), (240, 240), (500, 333)), ((40, 120, 120, 200), (240, 240), (17, 31)), ((1.5, 2.5, 3.5, 4.5), (10, 20), (20, 10))]
    assert [pi.rescale_box(b, f, t) for b, f, t in cases] == host["rescale_box"]
    assert [OP.rescale_box(b, f, t) for b, f, t in cases] == host["rescale_box"]
    for m, ref in host["process_patch_method"].items():
        _same_outcome(ref, _outcome(lambda: list(pi._process_patch_method(m))), m)
    for s, ref in host["str2bool"].items():
        assert pi.str2bool(s) == ref, s


def test_oracle_chunker_reproduces_the_references_crops(host, arrays):
    """reference chunk_image (PIL resize + crop) == oracle chunker (C restatement of Pillow's resampler): same boxes, same pixels"""
    imgs = RC.images()
    n_checked = 0
    for key, ref in host["chunk_image"].items():
        if ":" not in key or "raises" in ref:
            continue
        ii, method = key.split(":", 1)
        base, _, q = method.partition("?")
        params = dict(x.split("=") for x in q.split("&")) if q else {}
        patches, boxes = OP.chunk_image_simple(np.asarray(imgs[int(ii)]), int(params.get("hn", 3)), int(params.get("wn", 3)), base == "overlap")
        assert len(patches) == ref["n"], key
        assert [[float(v) for v in b] for b in boxes] == ref["boxes"], key
        assert [list(p.shape[1::-1]) for p in patches] == ref["sizes"], key
        assert [_sha(p) for p in patches] == ref["sha"], key
        n_checked += len(patches)
    assert n_checked > 250
    assert np.array_equal(arrays["chunk:2:simple:0"].shape, (240, 240, 3))


# ---- text ---------------------------------------------------------------------------------------------------------------
def test_split_text_matches_reference(host):
    from marqo_amd.s2_inference.processing import text as pt
    st = host["split_text"]
    for by, n, ov in RC.SPLIT_CASES:
        _same_outcome(st[f"{by}/{n}/{ov}"], _outcome(lambda: pt.split_text(RC.SPLIT_TEXT, split_by=by, split_length=n, split_overlap=ov)), (by, n, ov))
    for t in RC.SPLIT_EDGE_TEXTS:
        for by in ("sentence", "word", "character", "passage"):
            _same_outcome(st[f"edge:{t!r}/{by}"], _outcome(lambda: pt.split_text(t, split_by=by, split_length=2, split_overlap=1)), (t, by))
    _same_outcome(st["zero_length"], _outcome(lambda: pt.split_text("abc def", split_by="word", split_length=0, split_overlap=0)))
    _same_outcome(st["custom_sep"], _outcome(lambda: pt.split_text("a b c d e", split_by="word", split_length=2, split_overlap=0, custom_seperator="|")))
    _same_outcome(st["bad_split_by"], _outcome(lambda: pt.split_text("a b c", split_by="paragraphs")))
    _same_outcome(st["non_str_split_by"], _outcome(lambda: pt.split_text("a b c", split_by=3)))
    assert pt.prefix_text_chunks(["a", "b c"], "passage: ") == host["prefix_text_chunks"]["passage"]
    assert pt.prefix_text_chunks(["a"], "") == host["prefix_text_chunks"]["empty"]
    assert pt.prefix_text_chunks(["a"], None) == host["prefix_text_chunks"]["none"]
    for t in ["", " ", None, [], "x", "  \n", 3]:
        _same_outcome(host["check_make_string_valid"][repr(t)], _outcome(lambda: pt.check_make_string_valid(t)), repr(t))


# ---- the two wrappers: the reference's pooling / normalise / dispatch code over the oracle towers -------------------------------
def test_oracle_restates_the_hf_wrapper(arrays):
    cfg = RC.tiny_bert_cfg()
    sd = O.synthetic_bert_state_dict(cfg, seed=11)
    ids, mask = torch.from_numpy(arrays["hf:input_ids"]), torch.from_numpy(arrays["hf:attention_mask"])
    # the product's own WordPiece tokeniser gives the ids transformers' BertTokenizer gave inside the reference wrapper
    from marqo_amd.engine.tokenizers import WordPieceTokenizer
    tok = WordPieceTokenizer(RC.bert_vocab())(RC.WRAPPER_TEXTS, max_length=16)
    assert np.array_equal(tok["input_ids"], arrays["hf:input_ids"]) and np.array_equal(tok["attention_mask"], arrays["hf:attention_mask"])
    for pooling in ("mean", "cls"):
        c = O.BertConfig(**{**cfg.__dict__, "pooling": pooling})
        for norm in (True, False):
            ref = arrays[f"hf:{pooling}:{int(norm)}"]
            got = O.hf_encode(sd, c, ids, mask, normalize=norm).numpy()
            assert ref.dtype == np.float32 and ref.shape == (len(RC.WRAPPER_TEXTS), cfg.width)
            assert np.abs(got - ref).max() < 1e-6, (pooling, norm)
        one = WordPieceTokenizer(RC.bert_vocab())([RC.WRAPPER_TEXTS[1]], max_length=16)
        got1 = O.hf_encode(sd, c, torch.from_numpy(one["input_ids"]), torch.from_numpy(one["attention_mask"])).numpy()
        assert arrays[f"hf:{pooling}:str"].shape == (1, cfg.width) and np.abs(got1 - arrays[f"hf:{pooling}:str"]).max() < 1e-6


def test_oracle_restates_the_open_clip_wrapper(arrays):
    vcfg, tcfg = RC.TINY_VIT, RC.tiny_text_cfg()
    sd = O.synthetic_vit_state_dict(vcfg, seed=1)
    sd.update(O.synthetic_clip_text_state_dict(tcfg, seed=2))
    imgs = RC.images()
    # the oracle's Pillow-exact transform == the PIL transform the reference wrapper ran
    px = np.stack([OP.clip_transform(np.asarray(i), vcfg.image_size) for i in imgs])
    assert np.abs(px - arrays["clip:pixels"]).max() < 1e-6
    from marqo_amd.engine.tokenizers import ClipBpeTokenizer
    ids = ClipBpeTokenizer(RC.clip_merges(), context_length=77)(RC.WRAPPER_TEXTS)
    assert np.array_equal(ids, arrays["clip:ids"])
    for norm in (True, False):
        got = O.vit_forward(sd, vcfg, torch.from_numpy(px), normalize=norm).numpy()
        assert np.abs(got - arrays[f"clip:image:{int(norm)}"]).max() < 2e-6
        got = O.clip_text_forward(sd, tcfg, torch.from_numpy(ids), normalize=norm).numpy()
        assert np.abs(got - arrays[f"clip:text:{int(norm)}"]).max() < 2e-6
    full = arrays["clip:image:1"]
    assert np.abs(arrays["clip:image:single"] - full[1:2]).max() < 2e-6            # encode(x) == encode([x])
    assert np.abs(arrays["clip:image:tensors"] - full[:3]).max() < 2e-6            # pre-made tensors pass through un-reprocessed
    assert np.abs(arrays["clip:image:mixed"] - full[:3]).max() < 2e-6              # tensor | PIL | ndarray in one list
    assert np.abs(arrays["clip:encode:infer_image"] - full[:2]).max() < 2e-6       # dispatch: PIL -> image tower
    assert np.abs(arrays["clip:encode:infer_text"] - arrays["clip:text:1"][:2]).max() < 2e-6
    assert np.abs(arrays["clip:encode:default_image"] - full[:1]).max() < 2e-6
    assert arrays["clip:text:str"].shape == (1, vcfg.out_dim)


def test_product_dispatch_errors_match_reference(host):
    """abstract_clip_model.py:56-75: a `default` that is neither 'text' nor 'image' raises UnidentifiedImageError"""
    from marqo_amd.s2_inference.abstract_models import AbstractCLIPModel

    class _Fake(AbstractCLIPModel):
        def _load_necessary_components(self): pass
        def _check_loaded_components(self): pass
        def encode_text(self, inputs, normalize=True): return "text"
        def encode_image(self, inputs, normalize=True, image_download_headers=None): return "image"
    m = _Fake(device="cuda", model_properties={})
    with pytest.raises(UnidentifiedImageError):
        m.encode(["x"], default="audio", infer=False)
    assert host["clip_encode_bad_default"]["raises"] == "UnidentifiedImageError"
    assert m.encode(["a.jpg"], infer=True) == "image" and m.encode(["a.jpg is text"], infer=False) == "text"
    assert m.encode(["plain"], default="image", infer=False) == "image"


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference is not present on this machine")
def test_reference_callers_on_top_of_the_drop_in():
    """INTEGRATION.md §2 made executable: the reference's OWN caller classes (tensor_fields_container.py: SingleVectoriser,
    BatchCachingVectoriser, TensorFieldContent + TextChunker) with their `s2_inference` name bound to marqo_amd's module give the same
    embeddings as on the reference's own module (the CPU-runnable `random/small` model), and the product's errors travel through the
    reference's own `except` clauses to its ModelError — because, inside a host that has the `marqo` package, the product's error classes
    derive from the host's (marqo_amd/s2_inference/errors.py).  Also the search path (tensor_search.vectorise_jobs with enable_cache=True:
    same vectors, then served from the inference cache; unknown model -> the host's BadRequestError), model management (get_loaded_models
    parses the cache keys, eject_model's result dict, ModelNotInCacheError) and the index-settings model validation
    (core/models/marqo_index.py:150-200).  Runs in a fresh interpreter so that import order is the deployment's."""
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("MARQO_AMD_HOST_ERRORS", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_callers_check.py")], capture_output=True, text=True, env=env, timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stderr[-2000:]
    report = json.loads(lines[-1])
    assert r.returncode == 0 and report["mismatches"] == {}, report
    assert report["field_chunks"] == ["One sentence. Another one.", "A third. And a fourth."] and report["field_embeddings"] == [2, 32]


def test_error_classes_stand_alone_contract():
    """without a host application the error classes are plain: reference names, hierarchy, constructor, codes (api/exceptions.py:128-130,
    216-219,246-248)"""
    r = subprocess.run([sys.executable, "-c", (
        "import json; from marqo_amd.s2_inference import errors as E; c = E.ConfigurationError('bad'); u = E.UnknownModelError('m');"
        "print(json.dumps({'host': E._HOST_S2 is None and E._HOST_API is None, 'cfg': [c.code, int(c.status_code), c.message, isinstance(c, E.InternalError)],"
        "'mc': [E.ModelCacheManagementError('x').code, int(E.ModelCacheManagementError('x').status_code)], 'internal': E.InternalError(message='d').code,"
        "'s2': [u.message, str(u), isinstance(u, E.S2InferenceError), E.S2InferenceError().message]}))")],
        capture_output=True, text=True, timeout=120, cwd=ROOT, env=dict(os.environ, MARQO_AMD_HOST_ERRORS="0", PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out == {"host": True, "cfg": ["server_configuration_error", 500, "bad", True], "mc": ["model_cache_management_error", 409],
                   "internal": "internal", "s2": ["m", "m", True, None]}


# the reference's own unit-test files for this path that need neither network nor weights (the others load real checkpoints)
REFERENCE_TEST_FILES = {   # file under /root/reference/tests -> tests it holds
    "s2_inference/test_vectorise.py": 12,              # vectorise plumbing: batching by MARQO_MAX_VECTORISE_BATCH_SIZE, empty content, errors
    "s2_inference/test_encoding_random.py": 2,         # the `random` models through vectorise
    "s2_inference/test_sbert_utils.py": 2,             # Model / SBERT constructors without a device
    "core/inference/test_inference_cache.py": 14,      # MarqoInferenceCache: LRU / LFU, sizes, env validation, concurrency
    "core/inference/test_cache.py": 6,                 # MarqoLRUCache / MarqoLFUCache
    "processing/test_split_text.py": 8,                # split_text / prefix_text_chunks
    "core/inference/test_tensor_field_content.py": 21,      # the caller layer on top of the module: TensorFieldContent chunk / vectorise
    "core/inference/test_tensor_fields_container.py": 30,   # TensorFieldsContainer (field collection, multimodal sub-fields)
    "core/inference/test_tensor_field_vectorisers.py": 6,   # SingleVectoriser / BatchCachingVectoriser incl. the error mapping of every
                                                            # s2_inference error class (7 tests; one downloads images: deselected below)
    "core/inference/test_tensor_field_chunkers.py": 4,      # TextChunker (the product's split_text / prefixes underneath), AudioVideoChunker
                                                            # (8 tests; the four ImageChunker ones download their images: deselected below)
    "core/inference/test_vectorise_inference_cache.py": 9,  # vectorise(enable_cache=True) over the cache, incl. re-importing the module with
                                                            # other MARQO_INFERENCE_CACHE_* settings (10 tests; one loads a real CLIP on the CPU)
}
NEEDS_NETWORK = ["test_batch_vectoriser_should_support_different_content_chunk_types", "test_image_chunker", "test_vectorise_cacheWorkForImagePath"]


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference is not present on this machine")
def test_reference_unit_tests_pass_on_the_product():
    """The reference's OWN test files, run where they lie, with the module names they import and patch (`marqo.s2_inference.s2_inference`,
    `...random_utils`, `...sbert_utils`, `...processing.text`, `marqo.inference.inference_cache.*`) bound to marqo_amd's modules
    (tests/ref_suite_runner.py): every test they hold must pass on the product.  (Three interpreters side by side: the two cache files
    sleep through their expiry / concurrency cases for about a minute each.)"""
    import re
    slow = ("core/inference/test_inference_cache.py", "core/inference/test_vectorise_inference_cache.py")
    groups = [[f] for f in slow] + [[f for f in REFERENCE_TEST_FILES if f not in slow]]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("MARQO_AMD_HOST_ERRORS", None)
    deselect = " and ".join(f"not {n}" for n in NEEDS_NETWORK)
    def launch(g):
        files = [os.path.join(os.path.dirname(ref_shim.REFERENCE_SRC), "tests", f) for f in g]
        return subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ref_suite_runner.py"), *files, "-k", deselect],
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd="/tmp")

    def verdict(p):
        out, err = p.communicate(timeout=900)
        tail = out.strip().splitlines()[-1] if out.strip() else err[-2000:]
        m = re.search(r"(\d+) passed", tail)
        ok = p.returncode == 0 and m and "failed" not in tail and "error" not in tail
        return ok, (int(m.group(1)) if m else -1), out[-3000:] + err[-1500:]
    procs = [(g, launch(g)) for g in groups]
    for g, p in procs:
        ok, n, log = verdict(p)
        if not ok:   # the reference's cache tests race threads against sleeps: one more go on a loaded machine before calling it a failure
            ok, n, log = verdict(launch(g))
        assert ok, log
        assert n == sum(REFERENCE_TEST_FILES[f] for f in g), (g, n)


def test_product_segmenters_agree_with_the_independent_ones():
    """The reference-run fixtures above are produced with oracle/segment.py injected for nltk punkt (character scanners, written without
    looking at the product's regular expressions); the product's own rule-based splitters must agree with them on the fixture corpus AND on a
    seeded fuzz of the characters the rules speak about — two implementations of the same published boundary rules."""
    import random
    from marqo_amd.s2_inference.processing import text as pt
    from oracle import segment
    rng = random.Random(11)
    alphabet = "abc ABC .!? '\"()[]\n\t0 9dÉé東"
    corpus = list(RC.SPLIT_EDGE_TEXTS) + [RC.SPLIT_TEXT] + ["".join(rng.choice(alphabet) for _ in range(rng.randint(0, 80))) for _ in range(4000)]
    for t in corpus:
        assert pt._sentences(t) == segment.sentences(t), repr(t)
        assert pt._WORD.findall(t) == segment.words(t), repr(t)
    assert pt._sentences('He said "Go." Then left.') == ['He said "Go."', "Then left."]     # the closing quote stays with its sentence
