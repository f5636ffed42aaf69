"""Model registry + loader map of the engine.

Same shape as the reference's ``load_model_properties()`` (src/marqo/s2_inference/model_registry.py:2147-2187):
``{'models': {name: properties}, 'loaders': {type: loader class}}``.  Only the model families the gfx950 towers
run are registered: CLIP ViTs (B/32 … bigG/14) behind the ``open_clip`` (and OpenAI ``clip`` /
``fp16_clip`` naming) loaders, BERT-family encoders behind ``hf``, plus the reference's own plumbing fakes
``random`` (random_utils.py) and ``no_model``.  Entries are generated from the architecture tables in
``marqo_amd.engine.archs`` — names, dimensions, token limits and prefixes follow the reference registry
(model_registry.py:76-610 open_clip, :616-880 hf) so an index created against the reference resolves here.
"""
from __future__ import annotations

from typing import Dict

from marqo_amd.engine import archs

# pretrained tags the reference registers for the supported architectures (model_registry.py:76-610)
_OPEN_CLIP_TAGS = {
    # multilingual CLIPs: ViT image tower + XLM-RoBERTa text tower (open_clip HFTextEncoder; model_registry.py:262-273)
    "ViT-L-14-CLIPA-336": ("datacomp1b",),
    "roberta-ViT-B-32": ("laion2b_s12b_b32k",),
    "xlm-roberta-base-ViT-B-32": ("laion5b_s13b_b90k",),
    "xlm-roberta-large-ViT-H-14": ("frozen_laion5b_s13b_b90k",),
    "ViT-B-32": ["laion400m_e31", "laion400m_e32", "laion2b_e16", "laion2b_s34b_b79k", "openai"],
    "ViT-B-32-quickgelu": ["laion400m_e31", "laion400m_e32", "openai"],
    "ViT-B-32-256": ["datacomp_s34b_b86k"],
    "ViT-B-16": ["laion400m_e31", "laion400m_e32", "laion2b_s34b_b88k", "openai"],
    "ViT-B-16-quickgelu": ["metaclip_fullcc"],
    "ViT-B-16-plus-240": ["laion400m_e31", "laion400m_e32"],
    "ViT-L-14": ["laion400m_e31", "laion400m_e32", "laion2b_s32b_b82k", "openai"],
    "ViT-L-14-quickgelu": ["dfn2b"],
    "ViT-L-14-336": ["openai"],
    "ViT-H-14": ["laion2b_s32b_b79k"],
    "ViT-H-14-quickgelu": ["dfn5b"],
    "ViT-H-14-378-quickgelu": ["dfn5b"],
    "ViT-g-14": ["laion2b_s12b_b42k", "laion2b_s34b_b88k"],
    "ViT-bigG-14": ["laion2b_s39b_b160k"],
    # CoCa (model_registry.py:344-370): the contrastive towers of the captioner
    "coca_ViT-B-32": ["laion2b_s13b_b90k", "mscoco_finetuned_laion2b_s13b_b90k"],
    "coca_ViT-L-14": ["laion2b_s13b_b90k", "mscoco_finetuned_laion2b_s13b_b90k"],
    # EVA02-CLIP (model_registry.py:441-460): timm Eva trunks behind open_clip's TimmModel
    "EVA02-B-16": ["merged2b_s8b_b131k"], "EVA02-L-14": ["merged2b_s4b_b131k"], "EVA02-L-14-336": ["merged2b_s6b_b61k"],
    # SigLIP (model_registry.py:371-432)
    "ViT-B-16-SigLIP": ["webli"], "ViT-B-16-SigLIP-256": ["webli"], "ViT-B-16-SigLIP-384": ["webli"], "ViT-B-16-SigLIP-512": ["webli"],
    "ViT-L-16-SigLIP-256": ["webli"], "ViT-L-16-SigLIP-384": ["webli"], "ViT-SO400M-14-SigLIP-384": ["webli"],
}

# hf registry entries whose encoder the BERT tower runs (plain BERT: absolute positions, GELU, post-LN; XLM-RoBERTa checkpoints are
# the same encoder with shifted position ids; MPNet checkpoints add a relative-position attention bias): name -> (repo, dims, tokens, query prefix, chunk prefix, poolingMethod).  None = the
# reference entry has no such key (model_registry.py:616-880; tests/test_ref_parity.py compares every field with the reference's
# own dict).  The bge entries carry the reference's EXPLICIT "poolingMethod": "mean" (:804-849): the BAAI checkpoints ship a
# 1_Pooling/config.json that says CLS, and an index built with the reference is mean-pooled.
_BGE_EN = "Represent this sentence for searching relevant passages: "
_BGE_ZH = "为这个句子生成表示以用于检索相关文章："
_HF_BERT = {
    "hf/all-MiniLM-L6-v1": ("sentence-transformers/all-MiniLM-L6-v1", 384, 128, None, None, None),
    "hf/all-MiniLM-L6-v2": ("sentence-transformers/all-MiniLM-L6-v2", 384, 256, None, None, None),
    "hf/all-mpnet-base-v1": ("sentence-transformers/all-mpnet-base-v1", 768, 128, None, None, None),
    "hf/all-mpnet-base-v2": ("sentence-transformers/all-mpnet-base-v2", 768, 128, None, None, None),
    "hf/all_datasets_v3_mpnet-base": ("flax-sentence-embeddings/all_datasets_v3_mpnet-base", 768, 128, None, None, None),
    "hf/all_datasets_v4_mpnet-base": ("flax-sentence-embeddings/all_datasets_v4_mpnet-base", 768, 128, None, None, None),
    "hf/all_datasets_v3_MiniLM-L12": ("flax-sentence-embeddings/all_datasets_v3_MiniLM-L12", 384, 128, None, None, None),
    "hf/all_datasets_v3_MiniLM-L6": ("flax-sentence-embeddings/all_datasets_v3_MiniLM-L6", 384, 128, None, None, None),
    "hf/all_datasets_v4_MiniLM-L12": ("flax-sentence-embeddings/all_datasets_v4_MiniLM-L12", 384, 128, None, None, None),
    "hf/all_datasets_v4_MiniLM-L6": ("flax-sentence-embeddings/all_datasets_v4_MiniLM-L6", 384, 128, None, None, None),
    "hf/e5-small": ("intfloat/e5-small", 384, 192, "query: ", "passage: ", None),
    "hf/e5-base": ("intfloat/e5-base", 768, 192, "query: ", "passage: ", None),
    "hf/e5-large": ("intfloat/e5-large", 1024, 192, "query: ", "passage: ", None),
    "hf/e5-small-unsupervised": ("intfloat/e5-small-unsupervised", 384, 128, "query: ", "passage: ", None),
    "hf/e5-base-unsupervised": ("intfloat/e5-base-unsupervised", 768, 128, "query: ", "passage: ", None),
    "hf/e5-large-unsupervised": ("intfloat/e5-large-unsupervised", 1024, 128, "query: ", "passage: ", None),
    "hf/e5-small-v2": ("intfloat/e5-small-v2", 384, 512, "query: ", "passage: ", None),
    "hf/e5-base-v2": ("intfloat/e5-base-v2", 768, 512, "query: ", "passage: ", None),
    "hf/e5-large-v2": ("intfloat/e5-large-v2", 1024, 512, "query: ", "passage: ", None),
    "hf/multilingual-e5-small": ("intfloat/multilingual-e5-small", 384, 512, "query: ", "passage: ", None),
    "hf/multilingual-e5-base": ("intfloat/multilingual-e5-base", 768, 512, "query: ", "passage: ", None),
    "hf/multilingual-e5-large": ("intfloat/multilingual-e5-large", 1024, 512, "query: ", "passage: ", None),
    "hf/multilingual-e5-large-instruct": ("intfloat/multilingual-e5-large-instruct", 1024, 512,
                                          "Instruct: Given a web search query, retrieve relevant passages that answer the query\nQuery: ", None, None),
    "hf/bge-small-en-v1.5": ("BAAI/bge-small-en-v1.5", 384, 512, _BGE_EN, None, "mean"),
    "hf/bge-base-en-v1.5": ("BAAI/bge-base-en-v1.5", 768, 512, _BGE_EN, None, "mean"),
    "hf/bge-large-en-v1.5": ("BAAI/bge-large-en-v1.5", 1024, 512, _BGE_EN, None, "mean"),
    "hf/bge-small-zh-v1.5": ("BAAI/bge-small-zh-v1.5", 512, 512, _BGE_ZH, None, "mean"),
    "hf/bge-base-zh-v1.5": ("BAAI/bge-base-zh-v1.5", 768, 512, _BGE_ZH, None, "mean"),
    "hf/bge-large-zh-v1.5": ("BAAI/bge-large-zh-v1.5", 1024, 512, _BGE_ZH, None, "mean"),
    "hf/snowflake-arctic-embed-m": ("Snowflake/snowflake-arctic-embed-m", 768, 512, _BGE_EN, None, None),
    "hf/snowflake-arctic-embed-m-v1.5": ("Snowflake/snowflake-arctic-embed-m-v1.5", 768, 512, _BGE_EN, None, None),
    "hf/snowflake-arctic-embed-l": ("Snowflake/snowflake-arctic-embed-l", 1024, 512, _BGE_EN, None, None),
    "hf/ember-v1": ("llmrails/ember-v1", 1024, 512, None, None, None),
    "hf/GIST-large-Embedding-v0": ("avsolatorio/GIST-large-Embedding-v0", 1024, 512, None, None, None),
}


def _get_open_clip_properties() -> Dict:
    out = {}
    for arch_name, tags in _OPEN_CLIP_TAGS.items():
        vision, _ = archs.resolve_open_clip(arch_name)
        for tag in tags:
            name = f"open_clip/{arch_name}/{tag}"
            out[name] = {"name": name, "dimensions": vision.out_dim, "note": f"open_clip {arch_name} ({tag})",
                         "type": "open_clip", "pretrained": tag}
    for hub_name, arch_name in archs.KNOWN_HF_HUB_ARCHS.items():  # Marqo's fashion models (model_registry.py:483-494)
        vision, _ = archs.resolve_open_clip(arch_name)
        out[hub_name[len("hf-hub:"):]] = {"name": hub_name, "dimensions": vision.out_dim, "note": f"{hub_name} ({arch_name})", "type": "open_clip"}
    return out


_FP16_CLIP_NAMES = ("ViT-B/32", "ViT-B/16", "ViT-L/14")   # the reference registers no fp16/ViT-L/14@336px (model_registry.py:2069-2092)


def _get_clip_properties() -> Dict:
    """OpenAI-CLIP names (model_registry.py:16-73) and their fp16 variants (:2069-2092): same towers, QuickGELU."""
    out = {}
    for openai_name, arch_name in archs.OPENAI_CLIP_NAMES.items():
        vision, _ = archs.resolve_open_clip(arch_name, "openai")
        out[openai_name] = {"name": openai_name, "dimensions": vision.out_dim, "notes": "CLIP resnet", "type": "clip"}
        if openai_name in _FP16_CLIP_NAMES:
            out["fp16/" + openai_name] = {"name": "fp16/" + openai_name, "dimensions": vision.out_dim, "type": "fp16_clip",
                                          "notes": "reduced-precision CLIP; on MI355X every CLIP tower already runs bf16 MFMA"}
    return out


def _get_hf_properties() -> Dict:
    out = {}
    for key, (repo, dims, tokens, qp, cp, pooling) in _HF_BERT.items():
        p = {"name": repo, "dimensions": dims, "tokens": tokens, "type": "hf", "notes": ""}
        if qp is not None:
            p["text_query_prefix"] = qp
        if cp is not None:
            p["text_chunk_prefix"] = cp
        if pooling is not None:
            p["poolingMethod"] = pooling
        out[key] = p
    # the hf_stella entry (model_registry.py:898-904): Alibaba-NLP NewModel encoder, run natively (no remote code)
    out["Marqo/dunzhang-stella_en_400M_v5"] = {"name": "Marqo/dunzhang-stella_en_400M_v5", "dimensions": 1024, "tokens": 512,
                                              "type": "hf_stella", "trustRemoteCode": True}
    return out


# `sbert` entries (model_registry.py:538-613): SentenceTransformer checkpoints = an HF encoder + mean pooling (+ Normalize); every one of
# them also under its name without the organisation prefix (load_model_properties, :2149-2150)
_SBERT = {
    "sentence-transformers/all-MiniLM-L6-v1": (384, 128), "sentence-transformers/all-MiniLM-L6-v2": (384, 256),
    "sentence-transformers/all-MiniLM-L12-v2": (384, 256), "sentence-transformers/all-mpnet-base-v1": (768, 128),
    "sentence-transformers/all-mpnet-base-v2": (768, 128), "sentence-transformers/stsb-xlm-r-multilingual": (768, 128),
    "flax-sentence-embeddings/all_datasets_v3_MiniLM-L12": (384, 128), "flax-sentence-embeddings/all_datasets_v3_MiniLM-L6": (384, 128),
    "flax-sentence-embeddings/all_datasets_v4_MiniLM-L12": (384, 128), "flax-sentence-embeddings/all_datasets_v4_MiniLM-L6": (384, 128),
    "flax-sentence-embeddings/all_datasets_v3_mpnet-base": (768, 128), "flax-sentence-embeddings/all_datasets_v4_mpnet-base": (768, 128),
}


def _get_sbert_properties() -> Dict:
    out = {name: {"name": name, "dimensions": d, "tokens": t, "type": "sbert", "notes": ""} for name, (d, t) in _SBERT.items()}
    out.update({k.split("/")[-1]: v for k, v in list(out.items())})
    return out


def _get_sbert_test_properties() -> Dict:
    """the reference's plumbing entries (model_registry.py:976-999): all-MiniLM-L6-v1 truncated to 16 dimensions"""
    base = {"name": "sentence-transformers/all-MiniLM-L6-v1", "dimensions": 16, "tokens": 128, "type": "test", "notes": ""}
    return {"sentence-transformers/test": dict(base), "test": dict(base),
            "test_prefix": {**base, "text_query_prefix": "test query: ", "text_chunk_prefix": "test passage: "}}


def _get_sbert_onnx_properties() -> Dict:
    """`onnx/*` entries (model_registry.py:908-972): ONNX exports of the sbert checkpoints, served by the same HIP towers"""
    out = {}
    for repo, (d, t) in _SBERT.items():
        short = repo.split("/")[-1]
        if short in ("all-MiniLM-L12-v2", "stsb-xlm-r-multilingual"):   # (the reference registers no onnx export of these two)
            continue
        out["onnx/" + short] = {"name": repo, "dimensions": d, "tokens": t, "type": "sbert_onnx", "notes": ""}
    return out


# `onnx32/...` / `onnx16/...` entries (model_registry.py:1001-2065): ONNX exports of OpenAI / open_clip checkpoints; the ViT ones are served
# by the HIP towers of the checkpoint they were exported from.  model -> (dimensions, resolution)
_ONNX_CLIP = {
    "openai/ViT-L/14": (768, 224), "open_clip/ViT-L-14/openai": (768, 224), "open_clip/ViT-L-14/laion400m_e32": (768, 224),
    "open_clip/ViT-L-14/laion2b_s32b_b82k": (768, 224), "open_clip/ViT-L-14-336/openai": (768, 336),
    "open_clip/ViT-B-32/openai": (512, 224), "open_clip/ViT-B-32/laion400m_e31": (512, 224), "open_clip/ViT-B-32/laion400m_e32": (512, 224),
    "open_clip/ViT-B-32/laion2b_e16": (512, 224), "open_clip/ViT-B-32-quickgelu/openai": (512, 224),
    "open_clip/ViT-B-32-quickgelu/laion400m_e31": (512, 224), "open_clip/ViT-B-32-quickgelu/laion400m_e32": (512, 224),
    "open_clip/ViT-B-16/openai": (512, 224), "open_clip/ViT-B-16/laion400m_e31": (512, 224), "open_clip/ViT-B-16/laion400m_e32": (512, 224),
    "open_clip/ViT-B-16-plus-240/laion400m_e31": (640, 240), "open_clip/ViT-B-16-plus-240/laion400m_e32": (640, 240),
    "open_clip/ViT-H-14/laion2b_s32b_b79k": (1024, 224), "open_clip/ViT-g-14/laion2b_s12b_b42k": (1024, 224),
}


def _get_onnx_clip_properties() -> Dict:
    out = {}
    for model, (d, res) in _ONNX_CLIP.items():
        for prec, bits in (("onnx32", "float32"), ("onnx16", "float16")):
            name = f"{prec}/{model}"
            out[name] = {"name": name, "dimensions": d, "type": "clip_onnx", "resolution": res,
                         "note": f"the onnx {bits} export of {model}: served by the bf16 HIP towers of the same checkpoint"}
            if model.startswith("open_clip/"):
                # (metadata only; "laionb_s32b_b82k" sic: the spelling of the reference's own entry, model_registry.py:1090,1106)
                out[name]["pretrained"] = "laionb_s32b_b82k" if model == "open_clip/ViT-L-14/laion2b_s32b_b82k" else model.split("/")[2]
    return out


def _get_random_properties() -> Dict:
    """random/* plumbing fakes (model_registry.py:2094-2123)."""
    return {
        "random": {"name": "random", "dimensions": 384, "tokens": 128, "type": "random", "notes": ""},
        "random/large": {"name": "random/large", "dimensions": 768, "tokens": 128, "type": "random", "notes": ""},
        "random/small": {"name": "random/small", "dimensions": 32, "tokens": 128, "type": "random", "notes": ""},
        "random/medium": {"name": "random/medium", "dimensions": 128, "tokens": 128, "type": "random", "notes": ""},
    }


def _get_no_model_properties() -> Dict:
    return {"no_model": {"type": "no_model", "note": "special model no_model: users provide 'dimensions' and vectors"}}


def _get_model_load_mappings() -> Dict:
    # imported here so that `import marqo_amd.s2_inference.model_registry` stays cheap and free of cycles
    from marqo_amd.s2_inference.open_clip_model import CLIP, CLIP_ONNX, FP16_CLIP, MULTILINGUAL_CLIP, OPEN_CLIP
    from marqo_amd.s2_inference.hugging_face_model import HuggingFaceModel, HuggingFaceStellaModel
    from marqo_amd.s2_inference.random_utils import NO_MODEL, Random
    from marqo_amd.s2_inference.sbert_utils import SBERT, SBERT_ONNX, TEST
    return {"open_clip": OPEN_CLIP, "clip": CLIP, "fp16_clip": FP16_CLIP, "hf": HuggingFaceModel,
            "hf_stella": HuggingFaceStellaModel, "random": Random, "no_model": NO_MODEL, "sbert": SBERT, "test": TEST,
            "multilingual_clip": MULTILINGUAL_CLIP, "sbert_onnx": SBERT_ONNX, "clip_onnx": CLIP_ONNX}


def load_model_properties() -> Dict:
    models: Dict = {}
    models.update(_get_clip_properties())
    models.update(_get_sbert_properties())
    models.update(_get_sbert_test_properties())
    models.update(_get_sbert_onnx_properties())
    models.update(_get_onnx_clip_properties())
    models.update(_get_random_properties())
    models.update(_get_hf_properties())
    models.update(_get_open_clip_properties())
    models.update(_get_no_model_properties())
    from marqo_amd.s2_inference.open_clip_model import get_multilingual_clip_properties
    models.update(get_multilingual_clip_properties())
    return {"models": models, "loaders": dict(_get_model_load_mappings())}
