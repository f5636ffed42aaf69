"""hf_stella: the reference's HuggingFaceStellaModel (hugging_face_stella_model.py:9-23) is HuggingFaceModel over
Alibaba-NLP's custom `NewModel` encoder.  Host half: property validation exactly as the reference's own test
(tests/core/inference/embedding_models/test_hugging_face_stella_model.py:8-29), arch resolution from a NewModel config.json, the
rotary restatement pinned against transformers' rotate_half / apply_rotary_pos_emb.  GPU half: a tiny NewModel checkpoint in the
real on-disk format through vectorise() against the fp32 oracle (oracle/towers.py::new_model_encode)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import towers as O


def test_trust_remote_code_validation_like_the_reference():
    from marqo_amd.s2_inference.errors import InvalidModelPropertiesError
    from marqo_amd.s2_inference.hugging_face_model import HuggingFaceStellaModel
    for trc in (None, False):
        props = {k: v for k, v in {"name": "my_model", "type": "hf", "dimensions": 512, "trustRemoteCode": trc}.items() if v is not None}
        with pytest.raises(InvalidModelPropertiesError, match="trustRemoteCode"):
            HuggingFaceStellaModel(props, "cpu")
    assert HuggingFaceStellaModel({"name": "my_model", "type": "hf", "dimensions": 512, "trustRemoteCode": True}, "cpu") is not None


def test_registry_entry_and_arch():
    from marqo_amd.engine import archs
    from marqo_amd.s2_inference import s2_inference as s2
    p = s2.get_model_properties_from_registry("Marqo/dunzhang-stella_en_400M_v5")
    assert p == {"name": "Marqo/dunzhang-stella_en_400M_v5", "dimensions": 1024, "tokens": 512, "type": "hf_stella", "trustRemoteCode": True}
    a = archs.bert_arch_from_hf_config({"model_type": "new", "vocab_size": 30528, "max_position_embeddings": 8192, "hidden_size": 1024,
                                        "num_hidden_layers": 24, "num_attention_heads": 16, "intermediate_size": 4096, "hidden_act": "gelu",
                                        "layer_norm_eps": 1e-12, "position_embedding_type": "rope", "rope_theta": 160000,
                                        "rope_scaling": {"factor": 2.0, "type": "ntk"}, "type_vocab_size": 2, "pack_qkv": True})
    assert a == archs.STELLA_EN_400M and a.glu and a.rope_theta == 160000.0
    inv = a.rope_inv_freq()
    assert inv.shape == (32,) and torch.allclose(inv, O.new_model_inv_freq(O.NewModelConfig()))
    assert abs(float(inv[0]) - 2.0 ** (-2 / 64)) < 1e-6            # NTK: inv_freq[0] = 1 / factor^(2/d)
    with pytest.raises(KeyError):
        archs.bert_arch_from_hf_config({"model_type": "new", "vocab_size": 10, "max_position_embeddings": 8, "hidden_size": 64, "num_hidden_layers": 1,
                                        "num_attention_heads": 1, "intermediate_size": 64, "rope_scaling": {"type": "yarn", "factor": 2}})


def test_rotary_restatement_matches_transformers():
    from transformers.models.llama.modeling_llama import apply_rotary_pos_emb
    cfg = O.NewModelConfig(vocab=100, max_pos=64, width=128, layers=1, heads=2, mlp_dim=256)
    S = 13
    g = torch.Generator().manual_seed(0)
    q, k = torch.randn(2, 2, S, 64, generator=g), torch.randn(2, 2, S, 64, generator=g)
    freqs = torch.arange(S, dtype=torch.float32)[:, None] * O.new_model_inv_freq(cfg)[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    q1, k1 = apply_rotary_pos_emb(q, k, emb.cos()[None], emb.sin()[None])
    assert torch.equal(q1, q * emb.cos()[None, None] + O.rotate_half(q) * emb.sin()[None, None])
    assert torch.equal(k1, k * emb.cos()[None, None] + O.rotate_half(k) * emb.sin()[None, None])


def _write_tiny_new_model(root, name="tiny-stella"):
    from safetensors.torch import save_file
    from tests.test_tokenizers import _bert_vocab
    vocab = _bert_vocab()
    cfg = O.NewModelConfig(vocab=len(vocab), max_pos=128, width=128, layers=3, heads=2, mlp_dim=256, rope_theta=160000.0, rope_ntk_factor=2.0)
    d = root / "hf" / "acme" / name
    (d / "1_Pooling").mkdir(parents=True, exist_ok=True)
    sd = O.synthetic_new_model_state_dict(cfg, seed=7)
    save_file({"new." + k: v.contiguous() for k, v in sd.items()}, str(d / "model.safetensors"))
    (d / "config.json").write_text(json.dumps({
        "model_type": "new", "architectures": ["NewModel"], "vocab_size": cfg.vocab, "max_position_embeddings": cfg.max_pos, "hidden_size": cfg.width,
        "num_hidden_layers": cfg.layers, "num_attention_heads": cfg.heads, "intermediate_size": cfg.mlp_dim, "hidden_act": "gelu",
        "layer_norm_eps": 1e-12, "layer_norm_type": "layer_norm", "position_embedding_type": "rope", "rope_theta": 160000.0,
        "rope_scaling": {"factor": 2.0, "type": "ntk"}, "type_vocab_size": 2, "pack_qkv": True, "unpad_inputs": False,
        "use_memory_efficient_attention": False}))
    (d / "vocab.txt").write_text("\n".join(sorted(vocab, key=vocab.get)) + "\n")
    (d / "1_Pooling" / "config.json").write_text(json.dumps({"pooling_mode_cls_token": False, "pooling_mode_mean_tokens": True}))
    return {"name": f"acme/{name}", "dimensions": cfg.width, "tokens": 64, "type": "hf_stella", "trustRemoteCode": True}, sd, cfg, vocab


@pytest.mark.gpu
def test_stella_from_disk_vs_oracle(tmp_path):
    os.environ["MARQO_AMD_MODEL_DIR"] = str(tmp_path)
    os.environ["MARQO_MAX_CUDA_MODEL_MEMORY"] = "64"
    from marqo_amd.engine.tokenizers import WordPieceTokenizer
    from marqo_amd.s2_inference import s2_inference as s2
    from marqo_amd.s2_inference.errors import ModelLoadError
    try:
        s2.clear_loaded_models()
        props, sd, cfg, vocab = _write_tiny_new_model(tmp_path)
        texts = ["query: how much protein should a female eat", "the quick brown fox jumps over the lazy dog . " * 4, "a photo of a cat , a dog !", "fox"]
        out = np.asarray(s2.vectorise("tiny-stella", texts, model_properties=props, device="cuda:0"))
        tok = WordPieceTokenizer(vocab)(texts, max_length=64)
        ids, mask = torch.from_numpy(tok["input_ids"]), torch.from_numpy(tok["attention_mask"])
        ref = O.new_model_encode(sd, cfg, ids, mask).numpy()
        cos = (out * ref).sum(-1) / (np.linalg.norm(out, axis=-1) * np.linalg.norm(ref, axis=-1))
        print(f"NewModel (stella family) 3L: bf16 1-cos vs fp32 oracle {float((1 - cos).max()):.2e}")
        assert out.shape == (4, cfg.width) and float((1 - cos).max()) < 3e-4
        assert np.allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-5)
        # rotary positions matter: the same tokens shifted right by a prefix give a different vector than without rope would allow to tell;
        # here: batch composition / padding must NOT matter (positions are per sequence, padded keys masked)
        one = np.asarray(s2.vectorise("tiny-stella", [texts[3]], model_properties=props, device="cuda:0"))
        assert float(1 - (one[0] * out[3]).sum()) < 1e-5
        # the plain `hf` type refuses a custom-code checkpoint without trustRemoteCode, as the reference's AutoModel would
        with pytest.raises(ModelLoadError):
            s2.vectorise("tiny-stella-hf", texts, model_properties={"name": "acme/tiny-stella", "dimensions": cfg.width, "type": "hf"}, device="cuda:0")
    finally:
        s2.clear_loaded_models()
        os.environ.pop("MARQO_AMD_MODEL_DIR", None)


@pytest.mark.gpu
def test_stella_registry_size_synthetic_weights():
    """the registry entry at its real size (24 layers, width 1024, 30528 vocabulary) on seeded random weights: loads, runs, unit norm,
    and agrees with the fp32 oracle on the same weights"""
    os.environ["MARQO_AMD_SYNTHETIC_WEIGHTS"] = "1"
    os.environ["MARQO_MAX_CUDA_MODEL_MEMORY"] = "64"
    from marqo_amd.engine import archs, synthetic
    from marqo_amd.s2_inference import s2_inference as s2
    try:
        s2.clear_loaded_models()
        name = "Marqo/dunzhang-stella_en_400M_v5"
        texts = ["a short query", "a somewhat longer passage about tensor search engines and embeddings " * 3]
        out = np.asarray(s2.vectorise(name, texts, device="cuda:0"))
        assert out.shape == (2, 1024) and np.allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-5)
        model = s2.get_available_models()[s2._create_model_cache_key(name, "cuda:0", s2.get_model_properties_from_registry(name))]["model"]
        tok = model._tokenizer(texts, max_length=512)
        a = archs.STELLA_EN_400M
        sd = synthetic.random_bert_state_dict(a, seed=0)
        cfg = O.NewModelConfig(a.vocab, a.max_pos, a.width, a.layers, a.heads, a.mlp_dim, a.ln_eps, a.rope_theta, a.rope_ntk_factor)
        ref = O.new_model_encode(sd, cfg, torch.from_numpy(tok["input_ids"]), torch.from_numpy(tok["attention_mask"])).numpy()
        cos = (out * ref).sum(-1) / (np.linalg.norm(out, axis=-1) * np.linalg.norm(ref, axis=-1))
        print(f"stella_en_400M_v5 shape, 24L synthetic: bf16 1-cos vs fp32 oracle {float((1 - cos).max()):.2e}")
        assert float((1 - cos).max()) < 3e-4
    finally:
        s2.clear_loaded_models()
        os.environ.pop("MARQO_AMD_SYNTHETIC_WEIGHTS", None)
