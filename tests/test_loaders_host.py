"""Host-side contract of the loader types added on top of `open_clip` / `hf` (no GPU): constructors as the reference's, name resolution,
SentenceTransformer pipeline files, open_clip's HFTokenizer framing, registry / loader-map coverage."""
import json

import numpy as np
import pytest

from marqo_amd.s2_inference.errors import InternalError, InvalidModelPropertiesError
from marqo_amd.s2_inference.model_registry import load_model_properties


def test_loader_map_covers_the_reference_loader_types():
    reg = load_model_properties()
    with open(__file__.replace("test_loaders_host.py", "golden/ref_host.json"), encoding="utf-8") as f:
        ref_types = set(json.load(f)["registry_loader_types"])
    assert set(reg["loaders"]) == ref_types                      # (languagebind — video / audio — is a model type without a loader-map entry)
    by_type = {}
    for name, p in reg["models"].items():
        by_type.setdefault(p["type"], []).append(name)
        assert p["type"] in reg["loaders"], name
    assert len(by_type["sbert"]) == 24 and len(by_type["test"]) == 3 and len(by_type["sbert_onnx"]) == 10
    assert len(by_type["clip_onnx"]) == 38 and len(by_type["multilingual_clip"]) == 4
    assert all(n.split("/", 1)[0] in ("onnx16", "onnx32") for n in by_type["clip_onnx"])


def test_clip_onnx_names_resolve_to_the_exported_checkpoint():
    from marqo_amd.s2_inference.open_clip_model import CLIP_ONNX
    m = CLIP_ONNX("onnx16/openai/ViT-L/14", device="cuda", embedding_dim=768)
    assert m.model_properties.name == "open_clip/ViT-L-14/openai" and (m.onnx_type, m.source, m.clip_model) == ("onnx16", "openai", "ViT-L/14")
    m = CLIP_ONNX("onnx32/open_clip/ViT-B-16-plus-240/laion400m_e32", device="cuda", embedding_dim=640)
    assert m.model_properties.name == "open_clip/ViT-B-16-plus-240/laion400m_e32" and m.model_properties.dimensions == 640
    with pytest.raises(InternalError):
        CLIP_ONNX("onnx32/openai/ViT-L/14", device=None)
    for bad in ("onnx8/openai/ViT-L/14", "onnx32/somewhere/ViT-L-14", "onnx32", "onnx32/openai/RN50"):
        with pytest.raises(InvalidModelPropertiesError):
            CLIP_ONNX(bad, device="cuda", embedding_dim=512)


def test_multilingual_clip_constructor():
    from marqo_amd.s2_inference.open_clip_model import MULTILINGUAL_CLIP, get_multilingual_clip_properties
    table = get_multilingual_clip_properties()
    m = MULTILINGUAL_CLIP("multilingual-clip/XLM-R Large Vit-B/16+", device="cuda", embedding_dim=640)
    assert m.model_properties.name == "open_clip/ViT-B-16-plus-240/laion400m_e32" and m.textual_name == "M-CLIP/XLM-Roberta-Large-Vit-B-16Plus"
    m = MULTILINGUAL_CLIP("multilingual-clip/LABSE-Vit-L-14", device="cuda")
    assert m.model_properties.name == "open_clip/ViT-L-14/openai" and m.model_properties.dimensions == table[m.model_name]["dimensions"] == 768
    with pytest.raises(InternalError):
        MULTILINGUAL_CLIP("multilingual-clip/LABSE-Vit-L-14", device=None)
    with pytest.raises(InvalidModelPropertiesError):
        MULTILINGUAL_CLIP("multilingual-clip/nope", device="cuda")


def test_sentence_transformer_pipeline_files(tmp_path):
    from marqo_amd.s2_inference.sbert_utils import SBERT, SBERT_ONNX, TEST, _sentence_transformer_info
    assert _sentence_transformer_info(None) == (None, False) and _sentence_transformer_info(str(tmp_path)) == (None, False)
    (tmp_path / "sentence_bert_config.json").write_text(json.dumps({"max_seq_length": 384, "do_lower_case": False}))
    (tmp_path / "modules.json").write_text(json.dumps([{"idx": 0, "type": "sentence_transformers.models.Transformer"},
                                                        {"idx": 1, "type": "sentence_transformers.models.Pooling"},
                                                        {"idx": 2, "type": "sentence_transformers.models.Normalize"}]))
    assert _sentence_transformer_info(str(tmp_path)) == (384, True)
    (tmp_path / "modules.json").write_text("not json")
    assert _sentence_transformer_info(str(tmp_path)) == (384, False)
    for cls in (SBERT, TEST, SBERT_ONNX):
        with pytest.raises(InternalError):
            cls("sentence-transformers/all-MiniLM-L6-v2", device=None)
    t = TEST("sentence-transformers/all-MiniLM-L6-v1", device="cuda", embedding_dim=16, max_seq_length=128)
    assert t.truncated_embedding_dim == 16 and t.max_seq_length == 128 and t.model is None
    o = SBERT_ONNX("sentence-transformers/all-MiniLM-L6-v2", device="cuda", embedding_dim=384, cache_folder="x", onnx_folder="y", enable_overwrite=True)
    assert o.model_name_or_path == o.model_name == "sentence-transformers/all-MiniLM-L6-v2" and o.max_seq_length == 128


def test_open_clip_hf_tokenizer_framing():
    """open_clip's HFTokenizer: clean -> HF tokenizer (truncation to the context) -> ids padded to the context with <pad>"""
    from marqo_amd.s2_inference.open_clip_model import HfClipTokenizer

    class Fake:
        pad_id = 1

        def __call__(self, texts, max_length=None):
            enc = [[0] + [5 + len(w) for w in t.split()][:max_length - 2] + [2] for t in texts]
            S = max(len(e) for e in enc)
            ids = np.full((len(enc), S), 1, dtype=np.int64)
            for i, e in enumerate(enc):
                ids[i, :len(e)] = e
            return {"input_ids": ids}
    tok = HfClipTokenizer(Fake(), context_length=8)
    out = tok(["Tom &amp; Jerry   together ", "a b c d e f g h i j"])
    assert out.shape == (2, 8) and out.dtype == np.int64
    assert out[0].tolist() == [0, 8, 6, 10, 13, 2, 1, 1]            # "Tom & Jerry together": html-unescaped, whitespace collapsed
    assert out[1].tolist() == [0, 6, 6, 6, 6, 6, 6, 2]              # truncated to the context, </s> kept
    assert tok("x").shape == (1, 8)
