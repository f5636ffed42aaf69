"""Host logic of the vectorise() path, restated from the reference's own plumbing tests
(tests/s2_inference/test_vectorise.py, test_automatic_model_ejection_and_concurrency.py,
tests/core/inference/test_vectorise_inference_cache.py) against marqo_amd.s2_inference.  These use the reference's own
`random` fake model and mocks — no tower, no GPU."""
import datetime
import importlib
import os
import queue
import random
import threading
import time
from unittest import mock

import numpy as np
import pytest
from PIL import Image

from marqo_amd.s2_inference import random_utils, s2_inference
from marqo_amd.s2_inference.enums import AvailableModelsKey, Modality
from marqo_amd.s2_inference.errors import (ConfigurationError, InternalError, InvalidModelPropertiesError,
                                           ModelCacheManagementError, ModelLoadError, ModelNotInCacheError, S2InferenceError,
                                           UnknownModelError, VectoriseError)

S2 = "marqo_amd.s2_inference.s2_inference"


@pytest.fixture(autouse=True)
def _clean():
    s2_inference.clear_loaded_models()
    yield
    s2_inference.clear_loaded_models()


def _mock_model(dim=128):
    rnd = random_utils.Random(model_name="mock_model", embedding_dim=dim, device="cpu")
    m = mock.MagicMock()
    m.supports_dynamic_batching = False
    m.encode = mock.MagicMock(side_effect=lambda *a, **k: rnd.encode(*a, **k))
    props = {"name": "mock_model", "dimensions": dim, "tokens": 128, "type": "sbert"}
    avail = {s2_inference._create_model_cache_key("mock_model", "cpu", props): {
        AvailableModelsKey.model: m, AvailableModelsKey.model_size: 1,
        AvailableModelsKey.most_recently_used_time: datetime.datetime.now()}}
    return m, props, avail


def _vectorise_mock(m, props, avail, content, **kw):
    with mock.patch(S2 + "._available_models", avail), mock.patch(S2 + "._update_available_models", mock.MagicMock()):
        return s2_inference.vectorise(model_name="mock_model", content=content, model_properties=props, device="cpu", **kw)


# ---- tests/s2_inference/test_vectorise.py ------------------------------------------------------------------
def test_vectorise_in_batches():
    m, props, avail = _mock_model()
    out = _vectorise_mock(m, props, avail, ["just a single content"])
    assert len(out) == 1 and len(out[0]) == 128 and isinstance(out[0][0], float)


def test_vectorise_empty_content():
    m, props, avail = _mock_model()
    with pytest.raises(RuntimeError, match="(?i)empty list of batches"):
        _vectorise_mock(m, props, avail, [])


@pytest.mark.parametrize("batch_size,n", [(1, 5), (2, 5), (16, 33), (16, 16), (100, 7)])
def test_vectorise_in_batches_with_different_batch_sizes(batch_size, n):
    m, props, avail = _mock_model()
    content = [f"content {i}" for i in range(n)]
    with mock.patch.dict(os.environ, {"MARQO_MAX_VECTORISE_BATCH_SIZE": str(batch_size)}):
        out = _vectorise_mock(m, props, avail, content)
    assert len(out) == n
    assert m.encode.call_count == -(-n // batch_size)
    sizes = [len(c.args[0]) for c in m.encode.call_args_list]
    assert sizes == [batch_size] * (n // batch_size) + ([n % batch_size] if n % batch_size else [])


def test_vectorise_single_string_bypasses_batching():
    m, props, avail = _mock_model()
    out = _vectorise_mock(m, props, avail, "a bare string")
    assert len(out) == 1 and len(out[0]) == 128
    (args, kwargs) = m.encode.call_args
    assert args[0] == "a bare string" and kwargs["modality"] == Modality.TEXT and kwargs["normalize"] is True


def test_infer_kwarg_only_reaches_first_batch():
    """appendix A.2 quirk: `infer` is popped inside the batch loop"""
    m, props, avail = _mock_model()
    with mock.patch.dict(os.environ, {"MARQO_MAX_VECTORISE_BATCH_SIZE": "2"}):
        _vectorise_mock(m, props, avail, ["a", "b", "c"], infer=True)
    assert [c.kwargs["infer"] for c in m.encode.call_args_list] == [True, False]


def test_dynamic_batching_models_get_one_call():
    m, props, avail = _mock_model()
    m.supports_dynamic_batching = True
    out = _vectorise_mock(m, props, avail, [f"c{i}" for i in range(100)])
    assert len(out) == 100 and m.encode.call_count == 1


@pytest.mark.parametrize("bad", ["0", "-1", "abc", "1.5"])
def test__get_max_vectorise_batch_size_invalid(bad):
    with mock.patch.dict(os.environ, {"MARQO_MAX_VECTORISE_BATCH_SIZE": bad}):
        with pytest.raises(ConfigurationError):
            s2_inference._get_max_vectorise_batch_size()


def test__get_max_vectorise_batch_size_default_and_env():
    assert s2_inference._get_max_vectorise_batch_size() == 16
    with mock.patch.dict(os.environ, {"MARQO_MAX_VECTORISE_BATCH_SIZE": "7"}):
        assert s2_inference._get_max_vectorise_batch_size() == 7


def test_vectorise_with_no_device_fails():
    with pytest.raises(InternalError, match="cannot be called without setting device"):
        s2_inference.vectorise("random/small", "hello")
    with pytest.raises(InternalError):
        s2_inference.load_multimodal_model_and_get_preprocessors("random/small", {"type": "random", "dimensions": 32})


def test_vectorise_error_handling_image():
    m, props, avail = _mock_model()
    from PIL import UnidentifiedImageError
    m.encode.side_effect = UnidentifiedImageError("bad image")
    with pytest.raises(VectoriseError, match="Could not process given image"):
        _vectorise_mock(m, props, avail, ["http://x/y.jpg"])
    m.encode.side_effect = OSError("image file is truncated (3 bytes not processed)")
    with pytest.raises(VectoriseError):
        _vectorise_mock(m, props, avail, ["http://x/y.jpg"])
    m.encode.side_effect = OSError("disk on fire")
    with pytest.raises(OSError, match="disk on fire"):
        _vectorise_mock(m, props, avail, ["http://x/y.jpg"])


# ---- registry / validation ---------------------------------------------------------------------------------------
def test_registry_shape_and_known_entries():
    mp = s2_inference.MODEL_PROPERTIES
    assert set(mp) == {"models", "loaders"}
    assert {"open_clip", "clip", "fp16_clip", "hf", "hf_stella", "random", "no_model"} <= set(mp["loaders"])
    p = s2_inference.get_model_properties_from_registry("open_clip/ViT-B-32/laion2b_s34b_b79k")
    assert p["dimensions"] == 512 and p["type"] == "open_clip" and p["name"] == "open_clip/ViT-B-32/laion2b_s34b_b79k"
    assert s2_inference.get_model_properties_from_registry("open_clip/ViT-L-14/laion2b_s32b_b82k")["dimensions"] == 768
    e5 = s2_inference.get_model_properties_from_registry("hf/e5-base-v2")
    assert (e5["name"], e5["dimensions"], e5["tokens"], e5["type"]) == ("intfloat/e5-base-v2", 768, 512, "hf")
    assert e5["text_query_prefix"] == "query: " and e5["text_chunk_prefix"] == "passage: "
    with pytest.raises(UnknownModelError):
        s2_inference.get_model_properties_from_registry("definitely/not-a-model")
    # names and dimensions of the wider families follow the reference registry (model_registry.py:237-256, 371-432, 483-494)
    for name, dims in {"open_clip/ViT-H-14/laion2b_s32b_b79k": 1024, "open_clip/ViT-g-14/laion2b_s34b_b88k": 1024,
                       "open_clip/ViT-bigG-14/laion2b_s39b_b160k": 1280, "open_clip/ViT-H-14-378-quickgelu/dfn5b": 1024,
                       "open_clip/ViT-B-16-SigLIP-512/webli": 768, "open_clip/ViT-L-16-SigLIP-384/webli": 1024,
                       "open_clip/ViT-SO400M-14-SigLIP-384/webli": 1152, "Marqo/marqo-fashionSigLIP": 768, "Marqo/marqo-fashionCLIP": 512}.items():
        p = s2_inference.get_model_properties_from_registry(name)
        assert p["dimensions"] == dims and p["type"] == "open_clip", name
    assert s2_inference.get_model_properties_from_registry("Marqo/marqo-fashionSigLIP")["name"] == "hf-hub:Marqo/marqo-fashionSigLIP"


def test_validate_model_properties():
    v = s2_inference.validate_model_properties
    assert v("x", {"name": "n", "dimensions": 3})["type"] == "sbert"          # default type + tokens filled in
    assert v("x", {"name": "n", "dimensions": 3})["tokens"] == 128
    with pytest.raises(InvalidModelPropertiesError, match="missing key 'name'"):
        v("x", {"type": "open_clip", "dimensions": 512})
    with pytest.raises(InvalidModelPropertiesError, match="missing key 'dimensions'"):
        v("x", {"type": "hf", "name": "a/b"})
    with pytest.raises(InvalidModelPropertiesError, match="Invalid model type"):
        v("x", {"type": "banana", "name": "n", "dimensions": 3})
    for bad in (0, -1, 1.5, "512", None):
        with pytest.raises(InvalidModelPropertiesError, match="positive integer"):
            v("x", {"type": "hf", "name": "a/b", "dimensions": bad})
    with pytest.raises(InvalidModelPropertiesError, match="no_model"):
        v("not_no_model", {"type": "no_model", "dimensions": 3})
    assert v("no_model", {"type": "no_model", "dimensions": 3})["dimensions"] == 3
    assert issubclass(InvalidModelPropertiesError, S2InferenceError)


def test_model_cache_key_format():
    k = s2_inference._create_model_cache_key("m", "cuda:0", {"name": "n", "dimensions": 5, "type": "hf", "tokens": 7})
    assert k == "m||n||5||hf||7||cuda:0"
    assert s2_inference._create_model_cache_key("m", "cpu", None) == "m||||||||||cpu"


# ---- tests/s2_inference/test_automatic_model_ejection_and_concurrency.py -------------------------------------------
def test_get_model_size():
    g = s2_inference.get_model_size
    v = s2_inference.validate_model_properties
    for name, size in {"open_clip/ViT-L-14/openai": 1.5, "open_clip/ViT-L-14/laion400m_e31": 1.5,
                       "open_clip/ViT-B-16/laion2b_s34b_b88k": 1, "hf/e5-base-v2": 1, "random/small": 0.1}.items():
        assert g(name, v(name, None)) == size, name
    assert g("my_custom_clip", {"name": "ViT-L-14", "type": "open_clip", "dimensions": 768, "model_size": 1.53}) == 1.53
    assert g("my_custom_clip", {"name": "ViT-L/14", "dimensions": 768, "type": "clip"}) == 1.5
    assert g("whatever", {"name": "x", "type": "unknown", "dimensions": 1}) == 0.66


def test_thread_safe_function_guards():
    with pytest.raises(RuntimeError, match="thread safeness"):
        s2_inference._validate_model_into_device("m", {"type": "random"}, "cpu", calling_func="somebody")
    with pytest.raises(RuntimeError, match="threading safeness"):
        s2_inference._check_memory_threshold_for_model("cpu", 1, calling_func="somebody")
    with pytest.raises(RuntimeError, match="threading safeness"):
        s2_inference._load_model("random/small", {"type": "random", "dimensions": 32, "name": "random/small"}, "cpu", calling_func="x")
    m = s2_inference._load_model("random/small", {"type": "random", "dimensions": 32, "name": "random/small"}, "cpu", calling_func="unit_test")
    assert m.encode("hi").shape == (1, 32)


def test_check_memory_threshold_and_ejection():
    with mock.patch.dict(os.environ, {"MARQO_MAX_CPU_MODEL_MEMORY": "0.25"}):
        assert s2_inference._check_memory_threshold_for_model("cpu", 0.1, calling_func="unit_test") is True
        with pytest.raises(ModelCacheManagementError, match="larger than the device threshold"):
            s2_inference._check_memory_threshold_for_model("cpu", 0.3, calling_func="unit_test")
        with pytest.raises(ModelCacheManagementError, match="Unable to check the device cache"):
            s2_inference._check_memory_threshold_for_model("tpu", 0.1, calling_func="unit_test")
        # three 0.1 GB models under a 0.25 GB budget: the least recently used is ejected to make room
        for name in ("random/small", "random/medium", "random/large"):
            s2_inference.vectorise(name, "hello", device="cpu")
            time.sleep(0.01)
        keys = list(s2_inference.get_available_models())
        assert len(keys) == 2 and not any(k.startswith("random/small||") for k in keys)
        s2_inference.vectorise("random/medium", "renew medium", device="cpu")   # renew -> `large` is now the LRU
        s2_inference.vectorise("random/small", "hello", device="cpu")
        keys = list(s2_inference.get_available_models())
        assert len(keys) == 2 and not any(k.startswith("random/large||") for k in keys)


def test_eject_model_and_clear():
    s2_inference.vectorise("random/small", "hello", device="cpu")
    assert len(s2_inference.get_available_models()) == 1
    with pytest.raises(ModelNotInCacheError):
        s2_inference.eject_model("random/small", "cuda:3")
    assert s2_inference.eject_model("random/small", "cpu")["result"] == "success"
    assert len(s2_inference.get_available_models()) == 0
    with pytest.raises(ModelNotInCacheError):
        s2_inference.eject_model("random/small", "cpu")


def test_concurrent_first_load_is_rejected_not_queued():
    """s2_inference.py:293-297: while one thread loads, another thread needing a load gets ModelCacheManagementError;
    threads using an already cached model proceed."""
    started, release = threading.Event(), threading.Event()
    real_load = s2_inference._load_model

    def slow_load(*a, **k):
        started.set()
        release.wait(5)
        return real_load(*a, **k)

    q1, q2 = queue.Queue(), queue.Queue()

    def first():
        try:
            s2_inference.vectorise("random/small", "x", device="cpu"); q1.put("success")
        except Exception as e:  # pragma: no cover
            q1.put(e)

    def racer(name, q):
        try:
            s2_inference.vectorise(name, "x", device="cpu"); q.put("success")
        except Exception as e:
            q.put(e)

    s2_inference.vectorise("random/large", "warm", device="cpu")  # already cached model
    with mock.patch(S2 + "._load_model", slow_load):
        t = threading.Thread(target=first); t.start()
        assert started.wait(5)
        racers = [threading.Thread(target=racer, args=("random/medium", q2)) for _ in range(3)]
        cached = threading.Thread(target=racer, args=("random/large", q1))
        for r in racers + [cached]:
            r.start()
        for r in racers + [cached]:
            r.join()
        release.set(); t.join()
    assert sorted(str(q1.get()) for _ in range(2)) == ["success", "success"]
    assert q2.qsize() == 3
    while not q2.empty():
        assert isinstance(q2.get(), ModelCacheManagementError)


def test_model_load_error_wrapping():
    with pytest.raises(ModelLoadError, match="Unable to load model"):
        # cpu device -> the HIP engine refuses loudly (no CPU fallback), surfaced as ModelLoadError like any load failure
        s2_inference.vectorise("open_clip/ViT-B-32/laion2b_s34b_b79k", "hello", device="cpu")
    assert len(s2_inference.get_available_models()) == 0
    with pytest.raises(ModelLoadError):
        s2_inference.vectorise("hf/e5-base-v2", "hello", device="cpu")


def test_no_model_refuses_to_vectorise():
    with pytest.raises(VectoriseError, match="no_model"):
        s2_inference.vectorise("no_model", "hi", model_properties={"type": "no_model", "dimensions": 8}, device="cpu")


# ---- output conversion (s2_inference.py:623-749) -----------------------------------------------------------------------
def test_convert_vectorized_output():
    import torch
    c = s2_inference._convert_vectorized_output
    assert c(np.ones((2, 3), np.float32)) == [[1.0] * 3] * 2
    assert c(np.ones(3, np.float32)) == [[1.0] * 3]
    assert c(torch.ones(2, 3)) == [[1.0] * 3] * 2 and c(torch.ones(3)) == [[1.0] * 3]
    assert c([np.ones(3), np.zeros(3)]) == [[1.0] * 3, [0.0] * 3]
    assert c([[1.0, 2.0]]) == [[1.0, 2.0]]
    with pytest.raises(TypeError):
        c("nope")
    with pytest.raises(ValueError):
        c([])
    assert s2_inference._convert_cached_embeddings_to_output([1.0, 2.0]) == [[1.0, 2.0]]
    with pytest.raises(TypeError):
        s2_inference._convert_cached_embeddings_to_output(np.ones(3))


def test_vectorise_ndarray_fast_path():
    a = s2_inference.vectorise_ndarray("random/small", ["a", "b"], device="cpu")
    assert isinstance(a, np.ndarray) and a.shape == (2, 32)
    assert a.tolist() == s2_inference.vectorise("random/small", ["a", "b"], device="cpu")


# ---- inference cache (tests/core/inference/test_vectorise_inference_cache.py) ----------------------------------------------
@pytest.fixture
def cached_vectorise():
    with mock.patch.dict(os.environ, {"MARQO_INFERENCE_CACHE_SIZE": "50", "MARQO_INFERENCE_CACHE_TYPE": "LRU"}):
        importlib.reload(s2_inference)
        yield s2_inference.vectorise
    importlib.reload(s2_inference)


def test_cache_single_string(cached_vectorise):
    first = cached_vectorise("random/small", "test", device="cpu", enable_cache=True)
    with mock.patch(S2 + "._encode_without_cache") as enc:
        again = cached_vectorise("random/small", "test", device="cpu", enable_cache=True)
        enc.assert_not_called()
    assert again == first and isinstance(again[0], list)
    with mock.patch(S2 + "._encode_without_cache", return_value=[[0.0]]) as enc:
        cached_vectorise("random/small", "test", device="cpu", enable_cache=False)  # cache bypassed when not enabled per call
        enc.assert_called_once()


def test_cache_partial_list_only_misses_are_encoded(cached_vectorise):
    cached = ["test1", "test2"]
    cached_vectorise("random/small", cached, device="cpu", enable_cache=True)
    with mock.patch(S2 + "._encode_without_cache") as enc:
        cached_vectorise("random/small", cached + ["test3", "test4"], device="cpu", enable_cache=True)
        assert enc.call_args[0][1] == ["test3", "test4"]


def test_cache_partial_list_vectors_in_original_positions(cached_vectorise):
    rng = random.Random(0)
    for _ in range(5):
        s2_inference.clear_marqo_inference_cache()
        cached = [f"test{i}" for i in range(20)]
        # Random model seeds from the batch hash -> cache each item individually so vectors are per-item deterministic
        original = [cached_vectorise("random/small", c, device="cpu", enable_cache=True)[0] for c in cached]
        content = cached + [f"test{i}" for i in range(20, 40)]
        rng.shuffle(content)
        vectors = cached_vectorise("random/small", content, device="cpu", enable_cache=True)
        assert len(vectors) == 40
        assert [vectors[content.index(c)] for c in cached] == original
        rng.shuffle(content)
        with mock.patch(S2 + "._encode_without_cache") as enc:
            cached_vectorise("random/small", content, device="cpu", enable_cache=True)
            enc.assert_not_called()


def test_cache_does_not_serve_pil_images(cached_vectorise):
    content = [Image.fromarray(np.random.randint(0, 256, (8, 8, 3), dtype=np.uint8))]
    cached_vectorise("random/small", content, device="cpu", enable_cache=True)
    with mock.patch(S2 + "._encode_without_cache", return_value=[[0.0]]) as enc:
        cached_vectorise("random/small", content, device="cpu", enable_cache=True)
        enc.assert_called_once()


def test_cache_policies():
    from marqo_amd.s2_inference.inference_cache import EnvVarError, MarqoInferenceCache
    assert not MarqoInferenceCache(0).is_enabled()
    with pytest.raises(EnvVarError):
        MarqoInferenceCache(-1)
    with pytest.raises(EnvVarError):
        MarqoInferenceCache(3, "FIFO")
    lru = MarqoInferenceCache(2, "LRU")
    lru.set("k", "a", [1.0]); lru.set("k", "b", [2.0]); lru.get("k", "a"); lru.set("k", "c", [3.0])
    assert ("k", "a") in lru and ("k", "b") not in lru and lru.currsize == 2 and lru.maxsize == 2
    lfu = MarqoInferenceCache(2, "LFU")
    lfu.set("k", "a", [1.0]); lfu.set("k", "b", [2.0]); lfu.get("k", "a"); lfu.get("k", "a"); lfu.get("k", "b")
    lfu.set("k", "c", [3.0])
    assert ("k", "a") in lfu and ("k", "b") not in lfu and ("k", "c") in lfu
    with pytest.raises(TypeError):
        lru.get("k", 5)


# ---- input typing (image_download.py:28-71) ----------------------------------------------------------------------------
def test_is_image_rules(tmp_path):
    from PIL import UnidentifiedImageError
    from marqo_amd.s2_inference.image_input import _is_image
    import torch
    assert _is_image("photo.JPG") and _is_image(["a.png", "not looked at"]) and _is_image("https://example.com/x")
    assert not _is_image("just some text") and not _is_image(["text first.", "b.png"])
    assert _is_image(Image.new("RGB", (2, 2))) and _is_image(np.zeros((2, 2, 3))) and _is_image(torch.zeros(3, 2, 2))
    f = tmp_path / "file.txt"; f.write_text("x")
    with pytest.raises(UnidentifiedImageError):
        _is_image(str(f))
    with pytest.raises(UnidentifiedImageError):
        _is_image([])
    with pytest.raises(UnidentifiedImageError):
        _is_image(5)


# ---- text chunking (processing/text.py) -----------------------------------------------------------------------------------
def test_split_text_and_prefix():
    from marqo_amd.s2_inference.processing.text import prefix_text_chunks, split_text
    assert split_text("", "sentence") == [" "] and split_text("a", "word") == ["a"]
    assert split_text("abcdef", "character", 3, 1) == ["abc", "cde", "ef"]
    assert split_text("one two three four five", "word", 2, 0) == ["one two", "three four", "five"]
    assert split_text("p1\n\np2\n\np3", "passage", 1, 0) == ["p1", "p2", "p3"]
    assert split_text("Hello there. How are you? Fine.", "sentence", 2, 1) == ["Hello there. How are you?", "How are you? Fine."]
    with pytest.raises(ValueError):
        split_text("abc", "word", 0, 0)
    with pytest.raises(KeyError):
        split_text("abc def", "paragraphs")
    assert prefix_text_chunks(["a", "b"], "passage: ") == ["passage: a", "passage: b"]
    assert prefix_text_chunks(["a"], "") == ["a"] and prefix_text_chunks(["a"], None) == ["a"]


def test_chunk_image_method_parsing_and_boxes():
    from marqo_amd.s2_inference.processing import image as I
    assert I._process_patch_method("simple") == ("simple", {})
    assert I._process_patch_method("overlap?hn=3&wn=4") == ("overlap", {"hn": "3", "wn": "4"})
    assert len(I.generate_boxes((240, 240), 3, 3)) == 9 and len(I.generate_boxes((240, 240), 3, 3, True)) == 13
    img = Image.new("RGB", (40, 30))
    assert I.chunk_image(img, "cuda", None) == ([img], [(0, 0, 40, 30)])
    assert I.chunk_image("http://a/b.png", "cuda", "none") == (["http://a/b.png"], ["http://a/b.png"])
    with pytest.raises(ValueError):
        I.chunk_image(img, "cuda", "bogus")


def test_open_clip_architecture_resolution(tmp_path):
    """registry names, -quickgelu variants and hf-hub `open_clip_config.json` files -> tower architectures (no GPU involved):
    CLIP ViTs with any head width <= 128, SigLIP / timm trunks, and loud errors for what the engine does not run."""
    import json
    from marqo_amd.engine import archs as A
    from marqo_amd.s2_inference.errors import InvalidModelPropertiesError, ModelLoadError
    from marqo_amd.s2_inference.open_clip_model import OPEN_CLIP
    v, t = A.resolve_open_clip("ViT-H-14-378-quickgelu", "dfn5b")
    assert (v.tokens, v.width // v.heads, v.quick_gelu, t.width // t.heads) == (730, 80, True, 64)
    v, t = A.resolve_open_clip("ViT-SO400M-14-SigLIP-384")
    assert (v.tokens, v.pool, v.mlp_dim, v.out_dim, t.ctx, t.vocab, t.causal, t.proj_bias, t.prefix) == (729, "map", 4304, 1152, 64, 32000, False, True, "text.")
    with pytest.raises(KeyError):
        A.resolve_open_clip("RN50")
    m = OPEN_CLIP(device="cuda", model_properties={"name": "hf-hub:acme/x", "dimensions": 768, "type": "open_clip"})

    def cfg(d):
        (tmp_path / "open_clip_config.json").write_text(json.dumps({"model_cfg": d}))
        return m._resolve_archs("hf-hub:acme/x", None, str(tmp_path))
    # Marqo/marqo-fashionSigLIP style: timm trunk + custom text tower
    v, t = cfg({"embed_dim": 768, "custom_text": True,
                "vision_cfg": {"image_size": 224, "timm_model_name": "vit_base_patch16_siglip_224", "timm_pool": "map", "timm_proj": "none"},
                "text_cfg": {"context_length": 64, "vocab_size": 32000, "hf_tokenizer_name": "timm/ViT-B-16-SigLIP", "width": 768, "heads": 12,
                             "layers": 12, "no_causal_mask": True, "proj_bias": True, "pool_type": "last"}})
    assert (v.pool, v.tokens, v.width, v.ln_eps, t.causal, t.ctx, t.out_dim) == ("map", 196, 768, 1e-6, False, 64, 768)
    v, t = cfg({"embed_dim": 1024, "vision_cfg": {"image_size": 384, "timm_model_name": "vit_large_patch16_siglip_384", "timm_pool": "map", "timm_proj": "none"},
                "text_cfg": {"context_length": 64, "vocab_size": 32000, "width": 1024, "heads": 16, "layers": 24, "no_causal_mask": True, "proj_bias": True}})
    assert (v.tokens, v.width, v.layers, t.width) == (576, 1024, 24, 1024)
    # plain CLIP config with 80-wide heads (ViT-H class)
    v, t = cfg({"embed_dim": 1024, "vision_cfg": {"image_size": 224, "layers": 32, "width": 1280, "head_width": 80, "patch_size": 14},
                "text_cfg": {"context_length": 77, "vocab_size": 49408, "width": 1024, "heads": 16, "layers": 24}})
    assert (v.heads, v.mlp_dim, v.pool, t.causal) == (16, 5120, "cls", True)
    # not runnable: a timm trunk with a projection, an HF text tower, an embed_dim that is not the SigLIP width
    with pytest.raises(InvalidModelPropertiesError):
        cfg({"embed_dim": 768, "vision_cfg": {"timm_model_name": "vit_base_patch16_siglip_224", "timm_pool": "map", "timm_proj": "linear"}, "text_cfg": {}})
    with pytest.raises(InvalidModelPropertiesError):
        cfg({"embed_dim": 512, "vision_cfg": {"image_size": 224, "layers": 12, "width": 768, "patch_size": 16},
             "text_cfg": {"hf_model_name": "xlm-roberta-base", "hf_tokenizer_name": "xlm-roberta-base"}})
    with pytest.raises(InvalidModelPropertiesError):
        cfg({"embed_dim": 512, "vision_cfg": {"timm_model_name": "vit_base_patch16_siglip_224"}, "text_cfg": {"width": 768}})
    (tmp_path / "open_clip_config.json").unlink()
    with pytest.raises(ModelLoadError):
        m._resolve_archs("hf-hub:acme/x", None, str(tmp_path))
    assert m._resolve_archs("hf-hub:Marqo/marqo-fashionSigLIP", None, str(tmp_path))[0].pool == "map"  # known repo: table fallback


def test_bench_workload_table_resolves():
    """every bench.py workload names an architecture the engine resolves, with a positive algorithmic cost"""
    import importlib.util
    import os
    from marqo_amd.engine import archs as A
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert "vit_b32_image" in bench.WORKLOADS and bench.BF16_DENSE_PEAK_TFLOPS == 2500.0
    for name, wl in bench.WORKLOADS.items():
        if wl["kind"] == "stub":     # the launcher self-test (no GPU work, tests/test_bench_launcher.py)
            continue
        assert wl["kind"] in ("image", "clip_text", "bert", "mixed", "ingest", "chunked", "stream") and wl["batch"] > 0, name
        if wl["kind"] == "bert":
            assert A.HF_BERT_ARCHS[wl["arch"]].gflop_per_text(77) > 0
        else:
            v, t = A.resolve_open_clip(wl["arch"])
            assert v.gflop_per_image > 0 and t.gflop_per_text(t.ctx) > 0, name
    v, _ = A.resolve_open_clip("ViT-B-32")
    assert abs(v.gflop_per_image - 8.82) < 0.01          # the per-embedding figure BASELINE / SURVEY §8(d) quote
    v, _ = A.resolve_open_clip("ViT-L-14")
    assert abs(v.gflop_per_image - 162.0) < 0.1


def test_pil_pixels_zero_copy_view_matches_asarray():
    """engine/preprocess.pil_pixels: the Arrow view of a decoded Pillow image is its RGB bytes + one pad byte per pixel, for every size,
    for lazily opened files, and other modes fall back to the converted array."""
    import io
    from marqo_amd.engine import preprocess as P
    from marqo_amd.s2_inference.image_input import pil_to_pixels, pil_to_rgb_u8
    rng = np.random.default_rng(0)
    for h, w in [(224, 224), (1, 1), (3, 5), (1080, 1920), (333, 77)]:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        r = P.pil_pixels(Image.fromarray(a))
        got = r.view[..., :3] if isinstance(r, P.Rgbx) else r
        assert r.shape == (h, w, 3) and np.array_equal(got, a)
    b = io.BytesIO()
    Image.fromarray(rng.integers(0, 256, (100, 120, 3), dtype=np.uint8)).save(b, "PNG")
    lazy = Image.open(io.BytesIO(b.getvalue()))
    r = P.pil_pixels(lazy)
    assert np.array_equal(r.view[..., :3] if isinstance(r, P.Rgbx) else r, np.asarray(lazy))
    for mode in ("L", "RGBA", "P", "CMYK"):
        im = Image.fromarray(rng.integers(0, 256, (8, 9, 3), dtype=np.uint8)).convert(mode)
        r = pil_to_pixels(im)   # (palette sources arrive in the NEAREST-resize container; flatten_pixels is the plain RGB view of any container)
        got = r.view[..., :3] if isinstance(r, P.Rgbx) else P.flatten_pixels(r)
        assert np.array_equal(got, pil_to_rgb_u8(im))


def test_infer_modality_by_signature_and_extension():
    """multimodal_model_load.py:148-203 without python-magic: strings by URL-ness + the reference's extension lists (compared with the
    reference's own function in tests/ref_callers_check.py), bytes by container signature"""
    import io
    from PIL import Image as PILImage
    from marqo_amd.s2_inference import s2_inference as s2
    M = s2.Modality
    assert s2.infer_modality("hello") == M.TEXT and s2.infer_modality("photo.jpg") == M.TEXT       # not a URL: text, as in the reference
    assert s2.infer_modality("https://example.com/a.jpg") == M.IMAGE and s2.infer_modality("https://example.com/a.mp4") == M.VIDEO
    assert s2.infer_modality("https://example.com/a.wav") == M.AUDIO and s2.infer_modality([1, 2]) == M.TEXT
    for fmt in ("PNG", "JPEG", "GIF", "BMP", "WEBP", "TIFF"):
        b = io.BytesIO()
        PILImage.new("RGB", (4, 4), (10, 20, 30)).save(b, fmt)
        assert s2.infer_modality(b.getvalue()) == M.IMAGE, fmt
    assert s2.infer_modality(b"RIFF\x24\x00\x00\x00WAVEfmt ") == M.AUDIO and s2.infer_modality(b"RIFF\x24\x00\x00\x00AVI LIST") == M.VIDEO
    assert s2.infer_modality(b"\x00\x00\x00\x18ftypmp42\x00\x00") == M.VIDEO and s2.infer_modality(b"ID3\x03\x00") == M.AUDIO
    assert s2.infer_modality(b"OggS\x00\x02") == M.AUDIO and s2.infer_modality(b"just some text bytes") == M.TEXT and s2.infer_modality(b"") == M.TEXT
    assert s2.validate_url("https://example.com/x") and not s2.validate_url("not a url") and not s2.validate_url(3)
    assert s2.encode_url("https://example.com/ü b") == "https://example.com/%C3%BC%20b"


def test_infer_modality_probe_has_a_timeout_and_forwards_headers():
    """the extension-less-URL probe must not hang a request thread: (connect, read) timeout + the request's download headers;
    a timeout surfaces as MediaDownloadError"""
    import requests
    from marqo_amd.s2_inference.errors import MediaDownloadError
    seen = {}

    class Resp:
        def iter_content(self, chunk_size):
            yield b"\x89PNG\r\n\x1a\n" + b"0" * 100

        def close(self):
            pass

    def fake_get(url, **kw):
        seen.update(kw, url=url)
        return Resp()
    with mock.patch.object(requests, "get", fake_get):
        m = s2_inference.infer_modality("http://example.com/some/media", media_download_headers={"Authorization": "x"}, timeout_ms=1500)
    assert m == Modality.IMAGE and seen["timeout"] == (1.5, 1.5) and seen["headers"] == {"Authorization": "x"} and seen["stream"] is True

    def slow_get(url, **kw):
        raise requests.exceptions.ReadTimeout("too slow")
    with mock.patch.object(requests, "get", slow_get), pytest.raises(MediaDownloadError):
        s2_inference.infer_modality("http://example.com/some/media")


def test_staged_image_call_covers_every_image_once_in_equal_stages(monkeypatch):
    """open_clip_model._pipeline_stages: at least two stages, sizes within one image of each other apart from the last, contiguous, in order"""
    from marqo_amd.s2_inference import open_clip_model as M
    monkeypatch.setattr(M, "PIPELINE_CHUNK", 256)
    assert M._pipeline_stages(256) == [(0, 128), (128, 256)]
    assert M._pipeline_stages(512) == [(0, 256), (256, 512)]
    assert M._pipeline_stages(1024) == [(0, 256), (256, 512), (512, 768), (768, 1024)]
    assert [b - a for a, b in M._pipeline_stages(600)] == [300, 300]
    for n in (2, 3, 255, 257, 383, 385, 641, 1000, 4097):
        st = M._pipeline_stages(n)
        assert len(st) >= 2 and st[0][0] == 0 and st[-1][1] == n
        assert all(a1 == b0 for (_, b0), (a1, _) in zip(st[:-1], st[1:]))
        sizes = [b - a for a, b in st]
        assert len(set(sizes[:-1])) <= 1 and 0 < sizes[-1] <= sizes[0]
