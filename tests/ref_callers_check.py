"""Run by tests/test_ref_parity.py::test_reference_callers_on_top_of_the_drop_in in a FRESH interpreter (so that marqo_amd sees the host
package `marqo` at import time, as it would inside a Marqo deployment): the reference's own caller classes
(core/inference/tensor_fields_container.py: ModelConfig, SingleVectoriser, BatchCachingVectoriser, TensorFieldContent, TextChunker) with
ONE change — their module's `s2_inference` name bound to marqo_amd's — against the same classes on the reference's own s2_inference.
Prints one JSON object; exits non-zero on any mismatch.  Test infrastructure (imports oracle/ref_shim.py); never imported by the product."""
import json
import sys

import numpy as np


def main() -> int:
    from oracle import ref_shim
    from marqo_amd.s2_inference.processing import text as product_text   # (the sentence segmenter the reference would get from nltk's punkt
    ref_shim.install(sent_tokenize=product_text._sentences, word_tokenize=product_text._WORD.findall)   # data: not downloadable here)
    from marqo.core.inference import tensor_fields_container as T
    from marqo.core.exceptions import AddDocumentsError, ModelError
    from marqo.s2_inference import errors as host_errors
    from marqo.s2_inference.multimodal_model_load import Modality
    from marqo.tensor_search.telemetry import RequestMetricsStore
    from marqo.api import exceptions as host_api
    RequestMetricsStore.set_in_request(r=object())
    from marqo_amd.s2_inference import errors as our_errors
    from marqo_amd.s2_inference import s2_inference as ours
    report = {"host_errors_bound": our_errors._HOST_S2 is host_errors and our_errors._HOST_API is host_api}

    def cfg(name="random/small", normalize=True):
        return T.ModelConfig(model_name=name, model_properties=None, model_auth=None, device="cpu", normalize_embeddings=normalize)
    chunks = [("doc1_title_0", "hello world"), ("doc1_title_1", "second chunk"), ("doc2_body_0", "another document"), ("doc2_body_1", "")]
    ref_batch = T.BatchCachingVectoriser(Modality.TEXT, chunks, cfg())
    ref_single = T.SingleVectoriser(Modality.TEXT, cfg()).vectorise(["hello world", "second chunk"])
    ref_raw = T.SingleVectoriser(Modality.TEXT, cfg(normalize=False)).vectorise(["hello world"])

    T.s2_inference = ours          # <- the whole integration: the module the reference's callers call
    got_batch = T.BatchCachingVectoriser(Modality.TEXT, chunks, cfg())
    got_single = T.SingleVectoriser(Modality.TEXT, cfg()).vectorise(["hello world", "second chunk"])
    got_raw = T.SingleVectoriser(Modality.TEXT, cfg(normalize=False)).vectorise(["hello world"])
    report["batch_keys_equal"] = list(got_batch.embedding_cache) == list(ref_batch.embedding_cache)
    report["batch_embeddings_equal"] = all(np.array_equal(got_batch.embedding_cache[k], ref_batch.embedding_cache[k]) for k in ref_batch.embedding_cache)
    report["cached_lookup_equal"] = got_batch.vectorise(["x", "y"], "doc1_title") == ref_batch.vectorise(["x", "y"], "doc1_title")
    report["single_equal"] = np.array_equal(got_single, ref_single) and np.array_equal(got_raw, ref_raw)
    report["returns_lists_of_floats"] = isinstance(got_single, list) and isinstance(got_single[0], list) and isinstance(got_single[0][0], float)

    # TensorFieldContent.chunk / vectorise (the add_documents flow of one text field) on top of the drop-in
    from marqo.core.models.marqo_index import FieldType, TextPreProcessing, TextSplitMethod
    tp = TextPreProcessing(split_length=2, split_overlap=0, split_method=TextSplitMethod.Sentence)
    field = T.TensorFieldContent(field_content="One sentence. Another one. A third. And a fourth.", field_type=FieldType.Text, is_tensor_field=True)
    field.chunk({FieldType.Text: T.TextChunker(tp, "passage: ")})
    field.vectorise({FieldType.Text: T.SingleVectoriser(Modality.TEXT, cfg())})
    report["field_chunks"] = field.tensor_field_chunks
    report["field_embeddings"] = [len(field.tensor_field_embeddings), len(field.tensor_field_embeddings[0])]

    # error mapping through the reference's OWN except clauses (tensor_fields_container.py:155-163)
    def outcome(fn):
        try:
            fn()
            return "no error"
        except Exception as e:  # noqa: BLE001
            return f"{type(e).__module__}.{type(e).__name__}"
    report["unknown_model"] = outcome(lambda: T.SingleVectoriser(Modality.TEXT, cfg("no/such-model")).vectorise(["a"]))
    report["bad_properties"] = outcome(lambda: T.SingleVectoriser(Modality.TEXT, T.ModelConfig(
        model_name="my-model", model_properties={"dimensions": 8}, model_auth=None, device="cpu", normalize_embeddings=True)).vectorise(["a"]))
    report["empty_content"] = outcome(lambda: T.SingleVectoriser(Modality.TEXT, cfg()).vectorise([]))
    report["no_device"] = outcome(lambda: ours.vectorise("random/small", "a", device=None))
    # ---- the search path and model management (tensor_search/tensor_search.py:1876-1911, 2228-2244) --------------------------------
    from marqo.tensor_search import tensor_search as TS
    from marqo.tensor_search.models.search import VectorisedJobs
    from marqo.s2_inference import s2_inference as ref_s2

    def job(name="random/small", content=("what is marqo", "a second query")):
        props = ref_s2.get_model_properties_from_registry(name) if name == "random/small" else {"dimensions": 8}
        return VectorisedJobs(model_name=name, model_properties=props, content=list(content), device="cpu", normalize_embeddings=True,
                              image_download_headers=None, content_type="text", model_auth=None)
    TS.s2_inference = ref_s2
    ref_jobs = TS.vectorise_jobs([job()])
    TS.s2_inference = ours
    ours.clear_loaded_models()
    got_jobs = TS.vectorise_jobs([job()])
    report["search_jobs_equal"] = list(got_jobs) == list(ref_jobs) and all(
        list(got_jobs[k]) == list(ref_jobs[k]) and all(np.array_equal(got_jobs[k][c], ref_jobs[k][c]) for c in ref_jobs[k]) for k in ref_jobs)
    # enable_cache=True was passed: a second run must come from the inference cache with the same vectors
    report["search_jobs_cached_equal"] = all(np.array_equal(TS.vectorise_jobs([job()])[k][c], ref_jobs[k][c]) for k in ref_jobs for c in ref_jobs[k])
    report["search_unknown_model"] = outcome(lambda: TS.vectorise_jobs([job(name="no/such-model")]))
    loaded = TS.get_loaded_models()
    report["loaded_models"] = loaded
    report["eject"] = TS.eject_model("random/small", "cpu")
    report["loaded_after_eject"] = TS.get_loaded_models()
    report["eject_again"] = outcome(lambda: TS.eject_model("random/small", "cpu"))
    TS.s2_inference = ref_s2
    ref_s2.vectorise("random/small", "warm", device="cpu")
    report["ref_eject"] = TS.eject_model("random/small", "cpu")

    # ---- infer_modality, which tensor_search.py:74 and add_docs.py:24-25 import FROM the s2_inference module (no network: strings
    #      decided by URL-ness and extension, non-strings)
    from marqo.s2_inference.multimodal_model_load import infer_modality as ref_infer
    from PIL import Image as PILImage
    cases = ["a plain query", "", "photo.jpg", "/tmp/local/photo.png", "https://example.com/a.jpg", "https://example.com/a.JPEG",
             "http://example.com/path/b.png?x=1.webp", "https://example.com/v.mp4", "https://example.com/clip.MOV", "https://example.com/s.mp3",
             "https://example.com/s.ogg", "ftp://example.com/a.gif", "https://exa mple.com/a.jpg", "https://example.com/ü.png",
             "www.example.com/a.jpg", ["https://example.com/a.jpg"], PILImage.new("RGB", (2, 2)), 17, None]
    report["infer_modality_equal"] = [str(c)[:40] for c in cases if ours.infer_modality(c).value != ref_infer(c).value]

    # ---- image loading helpers the add_documents path takes from `clip_utils` (add_docs.py:23,139-143): same answers on local inputs
    import os as _os
    import tempfile
    from marqo.s2_inference import clip_utils as ref_clip
    from marqo_amd.s2_inference import clip_utils as our_clip
    d = tempfile.mkdtemp()
    png = _os.path.join(d, "a.png")
    PILImage.new("RGB", (5, 4), (1, 2, 3)).save(png)
    txt = _os.path.join(d, "a.txt")
    open(txt, "w").write("x")

    class Metrics:
        def __init__(self):
            self.log = []

        def start(self, k):
            self.log.append(("start", k))

        def stop(self, k):
            self.log.append(("stop", k))
    clip_diffs = []
    for thing in (png, txt, "https://example.com/a.jpg", "https://example.com/page", "plain text", "photo.bmp", [png], [], PILImage.new("RGB", (2, 2)), 3):
        a, b = outcome(lambda: str(ref_clip._is_image(thing))), outcome(lambda: str(our_clip._is_image(thing)))
        if a.split(".")[-1] != b.split(".")[-1]:
            clip_diffs.append(("_is_image", str(thing)[:30], a, b))
    ma, mb = Metrics(), Metrics()
    ia = ref_clip.load_image_from_path(png, {}, timeout_ms=1000, metrics_obj=ma)
    ib = our_clip.load_image_from_path(png, {}, timeout_ms=1000, metrics_obj=mb)
    if (ia.size, ia.mode, ma.log) != (ib.size, ib.mode, mb.log):
        clip_diffs.append(("load_image_from_path", ia.size, ib.size))
    for bad in ("not a path or url", _os.path.join(d, "missing.png")):
        a, b = outcome(lambda: ref_clip.load_image_from_path(bad, {})), outcome(lambda: our_clip.load_image_from_path(bad, {}))
        if a.split(".")[-1] != b.split(".")[-1]:
            clip_diffs.append(("load_image_from_path", bad, a, b))
    if sorted(ref_clip.get_allowed_image_types()) != sorted(our_clip.get_allowed_image_types()) or ref_clip.OPENAI_DATASET_MEAN != tuple(our_clip.OPENAI_DATASET_MEAN):
        clip_diffs.append(("constants",))
    report["clip_utils_diffs"] = clip_diffs

    # ---- every function the reference's s2_inference module defines exists here with the same parameter list
    import inspect
    api_missing, signature_diffs = [], []
    for fname, fn in vars(ref_s2).items():
        if inspect.isfunction(fn) and fn.__module__ == ref_s2.__name__:
            mine = getattr(ours, fname, None)
            if mine is None:
                api_missing.append(fname)
                continue
            shape = lambda f: [(p.name, str(p.kind), p.default is not inspect.Parameter.empty) for p in inspect.signature(f).parameters.values()]
            if shape(fn) != shape(mine):
                signature_diffs.append(fname)
    report["api_missing"], report["signature_diffs"] = sorted(api_missing), signature_diffs
    # ... and the loader classes of the loader map: every public method the reference's class has, ours has with the same leading
    # parameters (ours may append optional engine extensions such as return_device)
    import importlib
    pairs = [("marqo.core.inference.embedding_models.open_clip_model", "OPEN_CLIP", "marqo_amd.s2_inference.open_clip_model"),
             ("marqo.core.inference.embedding_models.hugging_face_model", "HuggingFaceModel", "marqo_amd.s2_inference.hugging_face_model"),
             ("marqo.s2_inference.clip_utils", "CLIP", "marqo_amd.s2_inference.open_clip_model"),
             ("marqo.s2_inference.clip_utils", "FP16_CLIP", "marqo_amd.s2_inference.open_clip_model"),
             ("marqo.s2_inference.clip_utils", "MULTILINGUAL_CLIP", "marqo_amd.s2_inference.open_clip_model"),
             ("marqo.s2_inference.random_utils", "Random", "marqo_amd.s2_inference.random_utils"),
             ("marqo.s2_inference.no_model_utils", "NO_MODEL", "marqo_amd.s2_inference.random_utils"),
             ("marqo.s2_inference.sbert_utils", "SBERT", "marqo_amd.s2_inference.sbert_utils"),
             ("marqo.s2_inference.sbert_utils", "TEST", "marqo_amd.s2_inference.sbert_utils"),
             ("marqo.s2_inference.sbert_onnx_utils", "SBERT_ONNX", "marqo_amd.s2_inference.sbert_utils"),
             ("marqo.inference.inference_cache.marqo_inference_cache", "MarqoInferenceCache", "marqo_amd.s2_inference.inference_cache")]
    internal = {"custom_clip_load", "extract_huggingface_archive", "mean_pooling"}   # download / onnxruntime helpers: not part of the path
    class_diffs = []
    for ref_mod, cls, our_mod in pairs:
        R, O = getattr(importlib.import_module(ref_mod), cls), getattr(importlib.import_module(our_mod), cls)
        for mname, fn in inspect.getmembers(R, inspect.isfunction):
            if (mname.startswith("_") and mname != "__init__") or mname in internal:
                continue
            mine = getattr(O, mname, None)
            if mine is None:
                if not (cls == "SBERT_ONNX" and mname == "normalize"):
                    class_diffs.append(f"{cls}.{mname}: missing")
                continue
            names = lambda f: [p.name for p in inspect.signature(f).parameters.values() if p.kind not in (p.VAR_POSITIONAL, p.VAR_KEYWORD)]
            rn, on = names(fn), names(mine)
            if mname == "__init__":   # constructors: the reference's named parameters must be accepted by name (ours may take **kwargs)
                accepts_kw = any(p.kind == p.VAR_KEYWORD for p in inspect.signature(mine).parameters.values())
                lost = [n for n in rn if n not in on and not accepts_kw]
                if lost:
                    class_diffs.append(f"{cls}.__init__ does not accept {lost}")
            elif on[:len(rn)] != rn:
                class_diffs.append(f"{cls}.{mname}: reference {rn}, product {on}")
    report["class_diffs"] = class_diffs

    # ---- index settings validation (core/models/marqo_index.py:150-200 calls validate_model_properties / get_model_properties_from_registry)
    from marqo.core.models import marqo_index as MI
    MI.s2_inference = ours
    m = MI.Model(name="hf/e5-base-v2")
    report["index_model_properties"] = {k: m.get_properties().get(k) for k in ("name", "dimensions", "type")}
    MI.s2_inference = ref_s2
    report["index_model_properties_ref"] = {k: MI.Model(name="hf/e5-base-v2").get_properties().get(k) for k in ("name", "dimensions", "type")}

    e = ours.errors.UnknownModelError("m") if hasattr(ours, "errors") else our_errors.UnknownModelError("m")
    report["is_host_class"] = (our_errors.UnknownModelError is host_errors.UnknownModelError and our_errors.InternalError is host_api.InternalError
                               and isinstance(e, host_errors.S2InferenceError) and ours.ModelDownloadError is host_errors.ModelDownloadError)
    expect = {"host_errors_bound": True, "batch_keys_equal": True, "batch_embeddings_equal": True, "cached_lookup_equal": True, "single_equal": True,
              "returns_lists_of_floats": True, "unknown_model": f"{ModelError.__module__}.ModelError",
              "bad_properties": f"{ModelError.__module__}.ModelError", "no_device": f"{host_api.InternalError.__module__}.InternalError",
              "is_host_class": True, "search_jobs_equal": True, "search_jobs_cached_equal": True,
              "search_unknown_model": f"{host_api.BadRequestError.__module__}.BadRequestError",
              "loaded_models": {"models": [{"model_name": "random/small", "model_device": "cpu"}]}, "loaded_after_eject": {"models": []},
              "eject": report["ref_eject"], "eject_again": f"{host_api.ModelNotInCacheError.__module__}.ModelNotInCacheError",
              "index_model_properties": report["index_model_properties_ref"], "infer_modality_equal": [], "clip_utils_diffs": [], "signature_diffs": [], "class_diffs": [],
              "api_missing": ["chunk_audio", "chunk_video", "load_multimodal_model"]}   # (the LanguageBind video / audio helpers: out of scope)
    bad = {k: (report.get(k), v) for k, v in expect.items() if report.get(k) != v}
    report["mismatches"] = {k: list(v) for k, v in bad.items()}
    print(json.dumps(report))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
