"""Generates the fixtures that pin parity to the REFERENCE ITSELF: this script imports the reference's own modules from
/root/reference/src (through oracle/ref_shim.py, which stubs the wheels that are not installed) and EXECUTES the functions
SURVEY.md §8(a) cites — it is the "outputs of the reference itself run here" anchor of the oracle.

    python tests/golden/make_ref_golden.py            # rewrite tests/golden/ref_host.json + ref_wrappers.npz
    python tests/golden/make_ref_golden.py --out DIR   # write elsewhere (tests/test_ref_parity.py re-runs it and diffs)

Reference functions executed (file:line under /root/reference/src/marqo):
  s2_inference/model_registry.py:2147-2187      load_model_properties  (the registry dict itself)
  s2_inference/s2_inference.py:48-158           vectorise / _encode_without_cache (with the reference's `random` models)
  s2_inference/s2_inference.py:239-283,340-407  _get_max_vectorise_batch_size, _create_model_cache_key, validate_model_properties
  s2_inference/s2_inference.py:503-517,623-749  get_model_size, _check_output_type, _convert_vectorized_output
  tensor_search/utils.py:334-340                generate_batches
  s2_inference/random_utils.py:11-64            sentence_to_hash, Random.encode
  core/inference/image_download.py:28-127       _is_image, format_and_load_CLIP_image(s)
  s2_inference/processing/image.py:46-151       chunk_image / PatchifySimple
  s2_inference/processing/image_utils.py:141-202,267-307  rescale_box, generate_boxes, patchify_image, _process_patch_method, str2bool
  s2_inference/processing/text.py:9-177         split_text, prefix_text_chunks, check_make_string_valid
  core/inference/embedding_models/hugging_face_model.py:172-214   HuggingFaceModel.encode, _average_pool_func, _cls_pool_func
  core/inference/embedding_models/open_clip_model.py:249-286 + abstract_clip_model.py:56-113   OPEN_CLIP.encode / encode_image /
                                                encode_text / _preprocess_images
  tensor_search/tensor_search.py:1940-1963 and core/.../tensor_fields_container.py:346-365 are pinned in tests/test_combine.py.

The towers inside the two wrappers are third-party code the reference imports (open_clip / transformers); here the reference's
wrapper objects get the fp32 oracle towers (oracle/towers.py) injected as `self.model` / `self._model`, `transformers`' own
BertTokenizer / CLIPTokenizer as the tokenizers, and a PIL transform equal to torchvision's Resize(bicubic)/CenterCrop/ToTensor/
Normalize as `self.preprocess` — so normalisation order, pooling, dispatch, batching and output conversion are REFERENCE code.
"""
import argparse
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402


def _sha(a) -> str:
    import numpy as np
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _exc(fn):
    try:
        return {"ok": fn()}
    except Exception as e:  # noqa: BLE001 - the exception TYPE is the known answer
        return {"raises": type(e).__name__, "bases": [c.__name__ for c in type(e).__mro__[1:-2]]}


def main(out_dir: str) -> None:
    # nltk punkt is not downloadable here: the reference's split_text runs on the INDEPENDENTLY written segmenters of oracle/segment.py (not
    # on the product's own, as in round 2 — that made the segmentation half of the comparison circular), so a fixture that the product
    # reproduces is two implementations of the documented boundary rules agreeing, plus the reference's windowing / re-joining code
    from oracle import segment
    ref_shim.install(sent_tokenize=segment.sentences, word_tokenize=segment.words)

    import numpy as np
    import torch
    from PIL import Image

    from oracle import preprocess as OP
    from oracle import towers as O
    from tests import ref_cases as RC

    os.environ.setdefault("MARQO_MAX_CPU_MODEL_MEMORY", "4")
    os.environ.setdefault("MARQO_MAX_CUDA_MODEL_MEMORY", "4")
    from marqo.s2_inference import s2_inference as ref_s2
    from marqo.s2_inference.model_registry import load_model_properties
    from marqo.s2_inference.random_utils import Random, sentence_to_hash
    from marqo.core.inference.image_download import _is_image, format_and_load_CLIP_image
    from marqo.s2_inference.processing import image as ref_image
    from marqo.s2_inference.processing import image_utils as ref_iu
    from marqo.s2_inference.processing import text as ref_text
    from marqo.tensor_search.utils import generate_batches
    from marqo.core.inference.embedding_models.hugging_face_model import HuggingFaceModel as RefHF
    from marqo.core.inference.embedding_models.open_clip_model import OPEN_CLIP as RefOpenClip

    host = {}
    arrays = {}

    # ---- registry --------------------------------------------------------------------------------------------------
    reg = load_model_properties()
    keep = ("name", "dimensions", "type", "tokens", "text_query_prefix", "text_chunk_prefix", "poolingMethod", "pretrained",
            "trustRemoteCode", "image_preprocessor", "imagePreprocessor", "tokenizer", "visual_model", "textual_model")
    host["registry_models"] = {k: {f: v[f] for f in keep if f in v} for k, v in sorted(reg["models"].items())
                               if v.get("type") in ("open_clip", "hf", "clip", "fp16_clip", "random", "no_model", "hf_stella", "sbert", "test", "multilingual_clip", "sbert_onnx", "clip_onnx")}
    host["registry_loader_types"] = sorted(reg["loaders"].keys())
    host["registry_all_types"] = {t: sum(1 for v in reg["models"].values() if v.get("type") == t)
                                  for t in sorted({v.get("type") for v in reg["models"].values()})}

    # ---- plumbing known answers -------------------------------------------------------------------------------------
    host["generate_batches"] = {f"{n}/{b}": [list(x) for x in generate_batches(list(range(n)), b)]
                                for n, b in [(0, 16), (1, 16), (16, 16), (17, 16), (40, 16), (5, 2), (5, 1)]}
    host["cache_key"] = [ref_s2._create_model_cache_key(n, d, p) for n, d, p in RC.CACHE_KEY_CASES]
    host["model_size"] = [ref_s2.get_model_size(n, p) for n, p in RC.MODEL_SIZE_CASES]
    conv = {}
    for name in RC.CONVERT_CASES:
        conv[name] = _exc(lambda: ref_s2._convert_vectorized_output(RC.convert_input(name)))
    host["convert_vectorized_output"] = conv
    host["check_output_type"] = {k: _exc(lambda v=v: ref_s2._check_output_type(v)) for k, v in {
        "ok": [[1.0, 2.0]], "int": [[1, 2]], "flat": [1.0, 2.0], "np_float": [[float(np.float32(1.5))]], "empty": []}.items()}
    bs = {}
    for val in (None, "1", "16", "64", "0", "-3", "abc", "2.5"):
        if val is None:
            os.environ.pop("MARQO_MAX_VECTORISE_BATCH_SIZE", None)
        else:
            os.environ["MARQO_MAX_VECTORISE_BATCH_SIZE"] = val
        bs[str(val)] = _exc(ref_s2._get_max_vectorise_batch_size)
    os.environ.pop("MARQO_MAX_VECTORISE_BATCH_SIZE", None)
    host["max_vectorise_batch_size"] = bs
    vp = {}
    for label, (name, props) in {
        "registry_name": ("hf/e5-base-v2", None), "unknown_no_props": ("not-a-model", None),
        "custom_hf": ("mine", {"name": "a/b", "dimensions": 8, "type": "hf"}),
        "custom_default_type": ("mine", {"name": "a/b", "dimensions": 8}),
        "missing_dims": ("mine", {"name": "a/b", "type": "hf"}), "missing_name": ("mine", {"dimensions": 8, "type": "sbert"}),
        "open_clip_localpath": ("mine", {"dimensions": 8, "type": "open_clip", "localpath": "/x"}),
        "no_model_ok": ("no_model", {"dimensions": 8, "type": "no_model"}), "no_model_no_dims": ("no_model", {"type": "no_model"}),
        "no_model_wrong_name": ("x", {"dimensions": 8, "type": "no_model"}), "bad_dims": ("mine", {"name": "q", "dimensions": -1, "type": "hf"}),
    }.items():
        vp[label] = _exc(lambda: ref_s2.validate_model_properties(name, None if props is None else dict(props)))
    host["validate_model_properties"] = vp

    # ---- the reference's random models: known answers through Random.encode and through the whole vectorise() ----------
    rnd = {"sentence_to_hash": {s: sentence_to_hash(s) for s in ["", "a", "hello", "hello world", "東京"]}}
    for name, dim in RC.RANDOM_CASES:
        m = Random(name, device="cpu", embedding_dim=dim)
        m.load()
        for i, inp in enumerate(RC.RANDOM_INPUTS):
            arrays[f"random:{name}:{i}"] = np.asarray(m.encode(inp), dtype=np.float64)
    host["random"] = rnd
    ref_s2.clear_loaded_models()
    vec = {}
    for label, kw in {
        "list": dict(model_name="random/small", content=["hello", "world", "marqo"], device="cpu"),
        "str": dict(model_name="random/small", content="hello", device="cpu"),
        "n40": dict(model_name="random/medium", content=[f"doc {i}" for i in range(40)], device="cpu"),
        "unnormalised": dict(model_name="random", content=["a", "b"], device="cpu", normalize_embeddings=False),
        "no_device": dict(model_name="random/small", content=["a"]),
        "empty_list": dict(model_name="random/small", content=[], device="cpu"),
        "unknown_model": dict(model_name="definitely/not-a-model", content=["a"], device="cpu"),
        "bad_props": dict(model_name="m", content=["a"], device="cpu", model_properties={"type": "random"}),
    }.items():
        r = _exc(lambda: ref_s2.vectorise(**kw))
        if "ok" in r:
            arrays[f"vectorise:{label}"] = np.asarray(r["ok"], dtype=np.float64)
            r = {"ok": {"n": len(r["ok"]), "d": len(r["ok"][0]), "elem_type": type(r["ok"][0][0]).__name__}}
        vec[label] = r
    host["vectorise_random"] = vec
    host["available_models_after"] = sorted(ref_s2.get_available_models().keys())
    ref_s2.clear_loaded_models()

    # ---- _is_image ----------------------------------------------------------------------------------------------------
    host["is_image"] = {label: _exc(lambda: bool(_is_image(RC.is_image_input(spec)))) for label, spec in RC.IS_IMAGE_CASES}
    fl = {}
    for label, spec in [("pil", ("pil", None)), ("ndarray", ("ndarray", None)), ("tensor", ("tensor", None)), ("int", ("int", 3))]:
        r = _exc(lambda: format_and_load_CLIP_image(RC.is_image_input(spec), {}))
        fl[label] = {"ok": type(r["ok"]).__name__ if not isinstance(r["ok"], Image.Image) else "PIL:" + r["ok"].mode} if "ok" in r else r
    host["format_and_load_CLIP_image"] = fl

    # ---- image chunking ------------------------------------------------------------------------------------------------
    host["generate_boxes"] = {f"{w}x{h}/{hn}x{wn}/{int(ov)}": [list(map(int, b)) for b in ref_iu.generate_boxes((w, h), hn, wn, overlap=ov)]
                              for (w, h) in [(240, 240), (100, 50), (17, 31), (7, 7)]
                              for (hn, wn) in [(3, 3), (2, 4), (1, 1), (5, 7)] for ov in (False, True) if h // hn and w // wn}
    host["rescale_box"] = [ref_iu.rescale_box(b, f, t) for b, f, t in [((0, 0, 80, 80), (240, 240), (500, 333)),
                                                                         ((40, 120, 120, 200), (240, 240), (17, 31)),
                                                                         ((1.5, 2.5, 3.5, 4.5), (10, 20), (20, 10))]]
    host["process_patch_method"] = {m: _exc(lambda: list(ref_iu._process_patch_method(m))) for m in
                                    RC.PATCH_METHODS + ["simple?hn", "overlap?hn=3&wn", "a/b?x=1&y=2"]}
    host["str2bool"] = {s: ref_iu.str2bool(s) for s in ["True", "true", "1", "t", "y", "yes", "False", "0", "no", "", "TRUE", "Yes"]}
    chunks = {}
    for ii, img in enumerate(RC.images()):
        for method in RC.PATCH_METHODS:
            key = f"{ii}:{method}"
            r = _exc(lambda: ref_image.chunk_image(img, "cpu", method))
            if "ok" in r:
                patches, boxes = r["ok"]
                chunks[key] = {"n": len(patches), "sizes": [list(p.size) for p in patches], "modes": [p.mode for p in patches],
                               "boxes": [[float(v) for v in b] for b in boxes],
                               "sha": [_sha(np.asarray(p)) for p in patches]}
                if ii in (2, 4) and method in ("simple", "overlap?hn=2&wn=2"):
                    for pi, p in enumerate(patches):
                        arrays[f"chunk:{ii}:{method}:{pi}"] = np.asarray(p)
            else:
                chunks[key] = r
    chunks["none_method_pil"] = [list(ref_image.chunk_image(RC.images()[0], "cpu", None)[1][0])]
    chunks["none_method_str"] = list(map(list, ref_image.chunk_image("some/path.jpg", "cpu", "")))
    chunks["bad_method"] = _exc(lambda: ref_image.chunk_image(RC.images()[0], "cpu", "not-a-method"))
    host["chunk_image"] = chunks

    # ---- text splitting ------------------------------------------------------------------------------------------------
    st = {}
    for by, n, ov in RC.SPLIT_CASES:
        st[f"{by}/{n}/{ov}"] = _exc(lambda: ref_text.split_text(RC.SPLIT_TEXT, split_by=by, split_length=n, split_overlap=ov))
    for t in RC.SPLIT_EDGE_TEXTS:
        for by in ("sentence", "word", "character", "passage"):
            st[f"edge:{t!r}/{by}"] = _exc(lambda: ref_text.split_text(t, split_by=by, split_length=2, split_overlap=1))
    st["zero_length"] = _exc(lambda: ref_text.split_text("abc def", split_by="word", split_length=0, split_overlap=0))
    st["custom_sep"] = _exc(lambda: ref_text.split_text("a b c d e", split_by="word", split_length=2, split_overlap=0, custom_seperator="|"))
    st["bad_split_by"] = _exc(lambda: ref_text.split_text("a b c", split_by="paragraphs"))
    st["non_str_split_by"] = _exc(lambda: ref_text.split_text("a b c", split_by=3))
    host["split_text"] = st
    host["prefix_text_chunks"] = {"passage": ref_text.prefix_text_chunks(["a", "b c"], "passage: "), "empty": ref_text.prefix_text_chunks(["a"], ""),
                                  "none": ref_text.prefix_text_chunks(["a"], None)}
    host["check_make_string_valid"] = {repr(t): _exc(lambda: ref_text.check_make_string_valid(t)) for t in ["", " ", None, [], "x", "  \n", 3]}

    # ---- HuggingFaceModel wrapper over the oracle BERT --------------------------------------------------------------------
    from transformers import BertTokenizer, CLIPTokenizer
    vocab = RC.bert_vocab()
    bcfg = RC.tiny_bert_cfg()
    bsd = O.synthetic_bert_state_dict(bcfg, seed=11)

    class _Out(tuple):
        last_hidden_state = None

    class _OracleBert:
        def __call__(self, input_ids=None, attention_mask=None, token_type_ids=None, **kw):
            h = O.bert_forward(bsd, bcfg, input_ids, attention_mask)
            o = _Out((h,))
            o.last_hidden_state = h
            return o

    for pooling in ("mean", "cls"):
        m = RefHF({"name": "acme/tiny-bert", "dimensions": bcfg.width, "tokens": 16, "type": "hf", "poolingMethod": pooling}, device="cpu")
        m._model, m._tokenizer = _OracleBert(), BertTokenizer(vocab=vocab, do_lower_case=True)
        m._pooling_func = m._load_pooling_method()
        m._check_loaded_components()
        for norm in (True, False):
            arrays[f"hf:{pooling}:{int(norm)}"] = m.encode(RC.WRAPPER_TEXTS, normalize=norm)
        arrays[f"hf:{pooling}:str"] = m.encode(RC.WRAPPER_TEXTS[1])
    tok = BertTokenizer(vocab=vocab, do_lower_case=True)(RC.WRAPPER_TEXTS, padding=True, truncation=True, max_length=16, return_tensors="np")
    arrays["hf:input_ids"], arrays["hf:attention_mask"] = tok["input_ids"], tok["attention_mask"]

    # ---- OPEN_CLIP wrapper over the oracle CLIP towers -----------------------------------------------------------------------
    merges = RC.clip_merges()
    vcfg, tcfg = RC.TINY_VIT, RC.tiny_text_cfg()
    csd = O.synthetic_vit_state_dict(vcfg, seed=1)
    csd.update(O.synthetic_clip_text_state_dict(tcfg, seed=2))

    class _OracleClip(torch.nn.Module):
        def encode_image(self, px):
            return O.vit_forward(csd, vcfg, px, normalize=False)

        def encode_text(self, ids):
            return O.clip_text_forward(csd, tcfg, ids, normalize=False)

    from marqo_amd.engine.tokenizers import ClipBpeTokenizer
    ours = ClipBpeTokenizer(merges, context_length=77)
    hf_vocab = {{"<start_of_text>": "<|startoftext|>", "<end_of_text>": "<|endoftext|>"}.get(t, t): i for t, i in ours.encoder.items()}
    hf_clip_tok = CLIPTokenizer(vocab=hf_vocab, merges=[tuple(x) for x in merges])

    def clip_tokenize(texts):  # what open_clip.tokenize returns: LongTensor [n, 77], SOT ... EOT then zeros, truncated with EOT last
        texts = [texts] if isinstance(texts, str) else texts
        out = torch.zeros(len(texts), 77, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = hf_clip_tok(t)["input_ids"]
            if len(ids) > 77:
                ids = ids[:77]
                ids[-1] = hf_clip_tok.eos_token_id
            out[i, :len(ids)] = torch.tensor(ids)
        return out

    def pil_transform(img):  # torchvision Resize(n_px, BICUBIC) -> CenterCrop -> convert RGB -> ToTensor -> Normalize, with PIL itself
        n = vcfg.image_size
        w, h = img.size
        short, long_ = (w, h) if w <= h else (h, w)
        new_short, new_long = n, int(n * long_ / short)
        nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
        img = img.resize((nw, nh), Image.BICUBIC)
        left, top = int(round((nw - n) / 2.0)), int(round((nh - n) / 2.0))
        img = img.crop((left, top, left + n, top + n)).convert("RGB")
        x = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float().div(255.0)
        mean = torch.tensor(OP.OPENAI_DATASET_MEAN).view(3, 1, 1)
        std = torch.tensor(OP.OPENAI_DATASET_STD).view(3, 1, 1)
        return (x - mean) / std

    oc = RefOpenClip(device="cpu", model_properties={"name": "open_clip/ViT-B-32/laion2b_s34b_b79k", "dimensions": vcfg.out_dim, "type": "open_clip"})
    oc.model, oc.tokenizer, oc.preprocess = _OracleClip(), clip_tokenize, pil_transform
    oc._check_loaded_components()
    imgs = RC.images()
    for norm in (True, False):
        arrays[f"clip:image:{int(norm)}"] = oc.encode_image(imgs, normalize=norm)
        arrays[f"clip:text:{int(norm)}"] = oc.encode_text(RC.WRAPPER_TEXTS, normalize=norm)
    arrays["clip:text:str"] = oc.encode_text(RC.WRAPPER_TEXTS[0])
    arrays["clip:image:single"] = oc.encode_image(imgs[1])
    arrays["clip:image:tensors"] = oc.encode_image([pil_transform(i) for i in imgs[:3]])
    arrays["clip:image:mixed"] = oc.encode_image([pil_transform(imgs[0]), imgs[1], np.asarray(imgs[2])])
    arrays["clip:encode:infer_image"] = oc.encode(imgs[:2])                                     # infer=True default, PIL -> image
    arrays["clip:encode:infer_text"] = oc.encode(RC.WRAPPER_TEXTS[:2])                          # plain strings -> text
    arrays["clip:encode:no_infer_default_text"] = oc.encode(["a.jpg is a file name"], infer=False)
    arrays["clip:encode:default_image"] = oc.encode(imgs[:1], default="image", infer=False)
    host["clip_encode_bad_default"] = _exc(lambda: oc.encode(["x"], default="audio", infer=False))
    arrays["clip:ids"] = clip_tokenize(RC.WRAPPER_TEXTS).numpy()
    arrays["clip:pixels"] = torch.stack([pil_transform(i) for i in imgs]).numpy()

    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "ref_host.json"), "w", encoding="utf-8") as f:
        json.dump(host, f, indent=1, sort_keys=True, ensure_ascii=False)
        f.write("\n")
    np.savez_compressed(os.path.join(out_dir, "ref_wrappers.npz"), **arrays)
    print(f"wrote {out_dir}/ref_host.json ({len(host)} sections) and ref_wrappers.npz ({len(arrays)} arrays)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE)
    main(ap.parse_args().out)
