"""End-to-end vectorise() through the engine loaders on the GPU, checked against the CPU oracle.

Mirrors the reference's invariants (tests/s2_inference/test_encoding.py: vectorise == model.encode, unit norm,
str == [str], dimensions) and the add_documents contract for `.preprocess` (test_add_documents_combined.py:411-439:
Tensor of shape (3, 224, 224)).  Real checkpoints do not exist offline: tiny checkpoints in the real on-disk formats
(open_clip safetensors + open_clip_config.json + BPE merges; HF config.json + model.safetensors + vocab.txt) are
written to a temp model dir, so the loaders' file handling is exercised, and registry-size models use seeded
random-init weights (MARQO_AMD_SYNTHETIC_WEIGHTS=1)."""
import gzip
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import preprocess as OP
from oracle import towers as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
COS_TOL = 1e-3


def _cos_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float((1 - (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))).max())


@pytest.fixture(scope="module")
def s2(tmp_path_factory):
    root = tmp_path_factory.mktemp("models")
    os.environ["MARQO_AMD_MODEL_DIR"] = str(root)
    os.environ["MARQO_MAX_CUDA_MODEL_MEMORY"] = "64"
    from marqo_amd.s2_inference import s2_inference
    s2_inference.clear_loaded_models()
    yield s2_inference, root
    s2_inference.clear_loaded_models()
    os.environ.pop("MARQO_AMD_MODEL_DIR", None)
    os.environ.pop("MARQO_AMD_SYNTHETIC_WEIGHTS", None)


# ---- tiny open_clip checkpoint on disk -----------------------------------------------------------------------------
def _tiny_clip(root):
    from safetensors.torch import save_file
    from tests.test_tokenizers import CORPUS, _train_bpe
    d = root / "hf-hub" / "acme" / "tiny-clip"
    d.mkdir(parents=True, exist_ok=True)
    merges = _train_bpe(CORPUS, 120)
    vocab = 512 + len(merges) + 2
    vcfg = O.VitConfig(image_size=64, patch_size=16, width=128, layers=2, heads=2, mlp_dim=256, out_dim=64)
    tcfg = O.ClipTextConfig(vocab=vocab, ctx=77, width=128, layers=2, heads=2, mlp_dim=256, out_dim=64)
    sd = O.synthetic_vit_state_dict(vcfg, seed=1)
    sd.update(O.synthetic_clip_text_state_dict(tcfg, seed=2))
    save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "open_clip_model.safetensors"))
    (d / "open_clip_config.json").write_text(json.dumps({"model_cfg": {
        "embed_dim": 64, "vision_cfg": {"image_size": 64, "layers": 2, "width": 128, "patch_size": 16, "head_width": 64, "mlp_ratio": 2.0},
        "text_cfg": {"context_length": 77, "vocab_size": vocab, "width": 128, "heads": 2, "layers": 2, "mlp_ratio": 2.0}}}))
    with gzip.open(d / "bpe_simple_vocab_16e6.txt.gz", "wt", encoding="utf-8") as f:
        f.write("#version: synthetic\n" + "\n".join(" ".join(m) for m in merges) + "\n")
    props = {"name": "hf-hub:acme/tiny-clip", "dimensions": 64, "type": "open_clip"}
    return props, sd, vcfg, tcfg


def test_open_clip_from_disk_text_and_image(s2):
    s2i, root = s2
    props, sd, vcfg, tcfg = _tiny_clip(root)
    texts = ["a photo of a cat", "the quick brown fox jumps over the lazy dog", "marqo is a tensor search engine"]
    out = s2i.vectorise("tiny-clip", texts, model_properties=props, device=DEV)
    assert len(out) == 3 and len(out[0]) == 64 and isinstance(out[0][0], float)
    model = s2i.get_available_models()[s2i._create_model_cache_key("tiny-clip", DEV, props)]["model"]
    ids = torch.from_numpy(model.tokenizer(texts))
    ref = O.clip_text_forward(sd, tcfg, ids).numpy()
    assert _cos_err(out, ref) < COS_TOL
    assert np.allclose(np.linalg.norm(np.asarray(out), axis=1), 1.0, atol=1e-5)
    # str == [str]; vectorise == model.encode
    assert _cos_err(s2i.vectorise("tiny-clip", texts[0], model_properties=props, device=DEV), out[:1]) < 3e-5   # one text = str, not list
    assert np.abs(model.encode(texts, normalize=True, infer=False) - np.asarray(out)).sum() < 1e-6
    # K14: the loader tokenises ASCII texts on the device; ids (hence embeddings) are identical to the host-tokeniser route,
    # also when a batch mixes in texts that must take the host route
    assert model._device_tokenizer is not None
    mixed = texts + ["naïve café über straße", "fish &amp; chips", "it's a dog's life"]
    via_host = model.text.encode_ids(torch.from_numpy(model.tokenizer(mixed))).cpu().numpy()
    assert np.array_equal(model.encode_text(mixed), via_host)
    # un-normalised
    raw = s2i.vectorise("tiny-clip", texts, model_properties=props, device=DEV, normalize_embeddings=False)
    assert not np.allclose(np.linalg.norm(np.asarray(raw), axis=1), 1.0, atol=1e-3)

    # images: PIL of arbitrary size -> GPU resize/crop -> tower; oracle = Pillow-exact CPU transform + fp32 tower
    rng = np.random.default_rng(0)
    pil = [Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for h, w in [(64, 64), (100, 80), (70, 200)]]
    emb = s2i.vectorise("tiny-clip", pil, model_properties=props, device=DEV, modality=s2i.Modality.IMAGE)
    px = torch.from_numpy(np.stack([OP.clip_transform(np.asarray(p), 64) for p in pil]))
    ref = O.vit_forward(sd, vcfg, px).numpy()
    assert _cos_err(emb, ref) < COS_TOL
    # `.preprocess` contract + encode_image on pre-made tensors gives the same embeddings
    model2, pre = s2i.load_multimodal_model_and_get_preprocessors("tiny-clip", props, DEV)
    assert model2 is model and pre["text"] is None
    t = pre["image"](pil[1])
    assert isinstance(t, torch.Tensor) and tuple(t.shape) == (3, 64, 64) and t.dtype == torch.float32
    assert torch.equal(t.cpu(), px[1]) or float((t.cpu() - px[1]).abs().max()) < 1e-6
    emb_t = s2i.vectorise("tiny-clip", [pre["image"](p).to(DEV) for p in pil], model_properties=props, device=DEV,
                          modality=s2i.Modality.IMAGE)
    assert _cos_err(emb_t, emb) < 1e-5
    # mixed list of tensors and PIL images
    emb_m = model.encode_image([pre["image"](pil[0]), pil[1], pil[2]])
    assert _cos_err(emb_m, emb) < 1e-5
    # image chunks on the device (K11) against the oracle chunker
    ce, boxes = model.encode_image_chunks([pil[2]], 3, 3, False)
    patches, bbs = OP.chunk_image_simple(np.asarray(pil[2]), 3, 3, False)
    refc = O.vit_forward(sd, vcfg, torch.from_numpy(np.stack([OP.clip_transform(p, 64) for p in patches]))).numpy()
    assert ce.shape == (1, 10, 64) and _cos_err(ce[0], refc) < COS_TOL and np.allclose(boxes[0], np.asarray(bbs), rtol=1e-6)


def test_chunk_image_reference_signature(s2):
    from marqo_amd.s2_inference.processing.image import chunk_image
    rng = np.random.default_rng(5)
    img = Image.fromarray(rng.integers(0, 256, (300, 400, 3), dtype=np.uint8))
    patches, boxes = chunk_image(img, DEV, "simple")
    ref_p, ref_b = OP.chunk_image_simple(np.asarray(img), 3, 3, False, backend="pil")
    assert len(patches) == 10 and all(isinstance(p, Image.Image) for p in patches)
    for p, r in zip(patches, ref_p):
        assert np.array_equal(np.asarray(p), r)
    assert np.allclose(np.asarray(boxes), np.asarray(ref_b))
    patches, _ = chunk_image(img, DEV, "overlap?hn=2&wn=2")
    assert len(patches) == 1 + 4 + 1


# ---- tiny HF BERT on disk ----------------------------------------------------------------------------------------------
def test_hf_from_disk(s2):
    s2i, root = s2
    from safetensors.torch import save_file
    from tests.test_tokenizers import _bert_vocab
    vocab = _bert_vocab()
    d = root / "hf" / "acme" / "tiny-bert"
    (d / "1_Pooling").mkdir(parents=True, exist_ok=True)
    cfg = O.BertConfig(vocab=len(vocab), max_pos=64, width=128, layers=2, heads=2, mlp_dim=256)
    sd = O.synthetic_bert_state_dict(cfg, seed=3)
    save_file({"bert." + k: v.contiguous() for k, v in sd.items()}, str(d / "model.safetensors"))  # HF checkpoints may carry the prefix
    (d / "config.json").write_text(json.dumps({"model_type": "bert", "vocab_size": len(vocab), "max_position_embeddings": 64,
                                               "hidden_size": 128, "num_hidden_layers": 2, "num_attention_heads": 2,
                                               "intermediate_size": 256, "hidden_act": "gelu", "layer_norm_eps": 1e-12}))
    (d / "vocab.txt").write_text("\n".join(sorted(vocab, key=vocab.get)) + "\n")
    (d / "1_Pooling" / "config.json").write_text(json.dumps({"pooling_mode_cls_token": False, "pooling_mode_mean_tokens": True}))
    props = {"name": "acme/tiny-bert", "dimensions": 128, "tokens": 32, "type": "hf"}
    texts = ["query: how much protein should a female eat", "the quick brown fox", "passage: a photo of a cat , a dog !", "fox"]
    out = s2i.vectorise("tiny-bert", texts, model_properties=props, device=DEV)
    from marqo_amd.engine.tokenizers import WordPieceTokenizer
    tok = WordPieceTokenizer(vocab)(texts, max_length=32)
    ref = O.hf_encode(sd, cfg, torch.from_numpy(tok["input_ids"]), torch.from_numpy(tok["attention_mask"])).numpy()
    assert np.asarray(out).shape == (4, 128) and _cos_err(out, ref) < COS_TOL
    # K14: device WordPiece == host WordPiece route, bit for bit, including texts that fall back to the host tokeniser
    model = s2i.get_available_models()[s2i._create_model_cache_key("tiny-bert", DEV, props)]["model"]
    assert model._device_tokenizer is not None
    mixed = texts + ["naïve café", "the fox [SEP] the dog", "", "word " * 100]
    th = model._tokenizer(mixed, max_length=32)
    via_host = model._model.encode_ids(torch.from_numpy(th["input_ids"]), torch.from_numpy(th["attention_mask"])).cpu().numpy()
    assert np.array_equal(model.encode(mixed), via_host)
    # padding invariance: a batch of one gives the same vector (appendix A.10)
    one = s2i.vectorise("tiny-bert", [texts[3]], model_properties=props, device=DEV)
    assert _cos_err(one, np.asarray(out)[3:]) < 1e-5
    # cls pooling via properties
    props_cls = dict(props, poolingMethod="cls")
    cfg.pooling = "cls"
    out_cls = s2i.vectorise("tiny-bert-cls", texts, model_properties=props_cls, device=DEV)
    ref_cls = O.hf_encode(sd, cfg, torch.from_numpy(tok["input_ids"]), torch.from_numpy(tok["attention_mask"])).numpy()
    assert _cos_err(out_cls, ref_cls) < COS_TOL
    # wrong dimensions -> load error
    from marqo_amd.s2_inference.errors import ModelLoadError
    with pytest.raises(ModelLoadError):
        s2i.vectorise("tiny-bert-bad", texts, model_properties=dict(props, dimensions=99), device=DEV)


def test_sbert_and_test_loader_types_from_disk(s2):
    """the reference's `sbert` / `test` loader types (sbert_utils.py:39-111) over a SentenceTransformer checkpoint directory: modules.json
    (Transformer -> Pooling -> Normalize), sentence_bert_config.json (max_seq_length), 1_Pooling; constructor and encode() contract of the
    reference; a pipeline that ends in Normalize returns unit vectors even for normalize=False; `test` truncates to 16 dimensions"""
    s2i, root = s2
    from safetensors.torch import save_file
    from marqo_amd.engine.tokenizers import WordPieceTokenizer
    from tests.test_tokenizers import _bert_vocab
    vocab = _bert_vocab()
    cfg = O.BertConfig(vocab=len(vocab), max_pos=64, width=128, layers=2, heads=2, mlp_dim=256)
    sd = O.synthetic_bert_state_dict(cfg, seed=7)
    texts = ["query: how much protein should a female eat", "the quick brown fox jumps over the lazy dog " * 3, "fox"]
    for repo, normalize_module in (("tiny-st", True), ("tiny-st-plain", False)):
        d = root / "hf" / "acme" / repo
        (d / "1_Pooling").mkdir(parents=True, exist_ok=True)
        save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "model.safetensors"))
        (d / "config.json").write_text(json.dumps({"model_type": "bert", "vocab_size": len(vocab), "max_position_embeddings": 64, "hidden_size": 128,
                                                   "num_hidden_layers": 2, "num_attention_heads": 2, "intermediate_size": 256,
                                                   "hidden_act": "gelu", "layer_norm_eps": 1e-12}))
        (d / "vocab.txt").write_text("\n".join(sorted(vocab, key=vocab.get)) + "\n")
        (d / "1_Pooling" / "config.json").write_text(json.dumps({"pooling_mode_cls_token": False, "pooling_mode_mean_tokens": True}))
        (d / "sentence_bert_config.json").write_text(json.dumps({"max_seq_length": 24, "do_lower_case": False}))
        mods = [{"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
                {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"}]
        if normalize_module:
            mods.append({"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"})
        (d / "modules.json").write_text(json.dumps(mods))
    tok = WordPieceTokenizer(vocab)
    # registry-style call: tokens from the properties
    props = {"name": "acme/tiny-st", "dimensions": 128, "tokens": 16, "type": "sbert"}
    out = np.asarray(s2i.vectorise("tiny-st", texts, model_properties=props, device=DEV))
    t = tok(texts, max_length=16)
    ref = O.hf_encode(sd, cfg, torch.from_numpy(t["input_ids"]), torch.from_numpy(t["attention_mask"])).numpy()
    assert out.shape == (3, 128) and _cos_err(out, ref) < COS_TOL
    raw = np.asarray(s2i.vectorise("tiny-st", texts, model_properties=props, device=DEV, normalize_embeddings=False))
    assert np.allclose(np.linalg.norm(raw, axis=1), 1, atol=1e-5)          # the checkpoint's own Normalize module
    # the loader class as the reference constructs it; max_seq_length from sentence_bert_config.json when none is given
    from marqo_amd.s2_inference.sbert_utils import SBERT, TEST
    m = SBERT("acme/tiny-st-plain", device=DEV, embedding_dim=128)
    m.load()
    assert m.max_seq_length == 24 and not m.always_normalized
    t24 = tok(texts, max_length=24)
    cfg_raw = O.hf_encode(sd, cfg, torch.from_numpy(t24["input_ids"]), torch.from_numpy(t24["attention_mask"]), normalize=False).numpy()
    got = m.encode(texts, normalize=False)
    assert isinstance(got, np.ndarray) and _cos_err(got, cfg_raw) < COS_TOL and not np.allclose(np.linalg.norm(got, axis=1), 1, atol=1e-3)
    with pytest.raises(Exception):
        SBERT("acme/tiny-st", device=None)
    # `test` type: the first 16 dimensions, normalised afterwards
    tprops = {"name": "acme/tiny-st-plain", "dimensions": 16, "tokens": 16, "type": "test"}
    t16 = np.asarray(s2i.vectorise("tiny-test", texts, model_properties=tprops, device=DEV))
    raw16 = O.hf_encode(sd, cfg, torch.from_numpy(t["input_ids"]), torch.from_numpy(t["attention_mask"]), normalize=False)[:, :16]
    assert t16.shape == (3, 16) and _cos_err(t16, torch.nn.functional.normalize(raw16, dim=1).numpy()) < COS_TOL
    assert isinstance(TEST("acme/tiny-st-plain", device=DEV).encode(texts[:1]), torch.Tensor)


# ---- registry-size models with synthetic weights (BASELINE configs 1-3) ---------------------------------------------------
def test_registry_models_synthetic_weights(s2):
    s2i, _ = s2
    from marqo_amd.s2_inference.errors import ModelLoadError
    with pytest.raises(ModelLoadError, match="(?s)no checkpoint.*MARQO_AMD_SYNTHETIC_WEIGHTS"):
        s2i.vectorise("open_clip/ViT-B-32/laion2b_s34b_b79k", "hello", device=DEV)
    os.environ["MARQO_AMD_SYNTHETIC_WEIGHTS"] = "1"
    # config 1: hf/e5-base-v2, batch = 8 short docs
    docs = [f"passage: synthetic document number {i} about topic {i % 3}" for i in range(8)]
    out = np.asarray(s2i.vectorise("hf/e5-base-v2", docs, device=DEV))
    assert out.shape == (8, 768) and np.allclose(np.linalg.norm(out, axis=1), 1, atol=1e-5)
    key = s2i._create_model_cache_key("hf/e5-base-v2", DEV, s2i.get_model_properties_from_registry("hf/e5-base-v2"))
    m = s2i.get_available_models()[key]["model"]
    tok = m._tokenizer(docs, max_length=512)
    from marqo_amd.engine import archs, synthetic
    sd = synthetic.random_bert_state_dict(archs.HF_BERT_ARCHS["intfloat/e5-base-v2"], seed=0)
    ref = O.hf_encode(sd, O.BertConfig(), torch.from_numpy(tok["input_ids"]), torch.from_numpy(tok["attention_mask"])).numpy()
    assert _cos_err(out, ref) < COS_TOL
    # config 2 through the API: ViT-B/32 on PIL images + text from the same model
    rng = np.random.default_rng(1)
    pil = [Image.fromarray(rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)) for _ in range(4)]
    name = "open_clip/ViT-B-32/laion2b_s34b_b79k"
    img = np.asarray(s2i.vectorise(name, pil, device=DEV, modality=s2i.Modality.IMAGE))
    txt = np.asarray(s2i.vectorise(name, ["a photo of a cat", "a dog"], device=DEV))
    assert img.shape == (4, 512) and txt.shape == (2, 512)
    varch, tarch = archs.resolve_open_clip("ViT-B-32")
    sdc = synthetic.random_open_clip_state_dict(vision=varch, text=tarch, seed=0)
    vc = O.VitConfig(224, 32, 768, 12, 12, 3072, 512)
    refi = O.vit_forward(sdc, vc, torch.from_numpy(np.stack([OP.clip_transform(np.asarray(p)) for p in pil]))).numpy()
    assert _cos_err(img, refi) < COS_TOL
    # CoCa through the API (registry name -> loader -> 76-position tokeniser + class embedding; attentional pooler on the image side)
    cname = "open_clip/coca_ViT-B-32/laion2b_s13b_b90k"
    cimg = np.asarray(s2i.vectorise(cname, pil, device=DEV, modality=s2i.Modality.IMAGE))
    ctxt = np.asarray(s2i.vectorise(cname, ["a photo of a cat", "a dog"], device=DEV))
    assert cimg.shape == (4, 512) and ctxt.shape == (2, 512) and np.allclose(np.linalg.norm(ctxt, axis=1), 1, atol=1e-5)
    cv, ct = archs.resolve_open_clip("coca_ViT-B-32")
    csd = synthetic.random_open_clip_state_dict(vision=cv, text=ct, seed=0)
    crefi = O.coca_vit_forward(csd, O.CocaVitConfig(224, 32, 768, 12, 12, 3072, 512), torch.from_numpy(np.stack([OP.clip_transform(np.asarray(p)) for p in pil]))).numpy()
    assert _cos_err(cimg, crefi) < COS_TOL
    ckey = s2i._create_model_cache_key(cname, DEV, s2i.get_model_properties_from_registry(cname))
    cm = s2i.get_available_models()[ckey]["model"]
    cids = torch.as_tensor(np.asarray(cm.tokenizer(["a photo of a cat", "a dog"])))
    assert cids.shape[1] == 76
    creft = O.coca_text_forward(csd, O.ClipTextConfig(ct.vocab, 77, ct.width, ct.layers, ct.heads, ct.mlp_dim, ct.out_dim), cids).numpy()
    assert _cos_err(ctxt, creft) < COS_TOL
    # EVA02-CLIP through the API (registry name -> loader -> timm Eva trunk under `visual.trunk.`, CLIP text tower under `text.`)
    ename = "open_clip/EVA02-B-16/merged2b_s8b_b131k"
    eimg = np.asarray(s2i.vectorise(ename, pil, device=DEV, modality=s2i.Modality.IMAGE))
    etxt = np.asarray(s2i.vectorise(ename, ["a photo of a cat", "a dog"], device=DEV))
    assert eimg.shape == (4, 512) and etxt.shape == (2, 512) and np.allclose(np.linalg.norm(eimg, axis=1), 1, atol=1e-5)
    ev, et = archs.resolve_open_clip("EVA02-B-16")
    esd = synthetic.random_open_clip_state_dict(vision=ev, text=et, seed=0)
    erefi = O.eva_vit_forward(esd, O.EvaVitConfig(224, 16, 768, 12, 12, 2048, 512), torch.from_numpy(np.stack([OP.clip_transform(np.asarray(p)) for p in pil]))).numpy()
    assert _cos_err(eimg, erefi) < COS_TOL
    ekey = s2i._create_model_cache_key(ename, DEV, s2i.get_model_properties_from_registry(ename))
    em = s2i.get_available_models()[ekey]["model"]
    eids = torch.as_tensor(np.asarray(em.tokenizer(["a photo of a cat", "a dog"])))
    ereft = O.clip_text_forward({k[len("text."):]: v for k, v in esd.items() if k.startswith("text.")},
                                O.ClipTextConfig(et.vocab, 77, et.width, et.layers, et.heads, et.mlp_dim, et.out_dim), eids).numpy()
    assert _cos_err(etxt, ereft) < COS_TOL
    s2i.eject_model(ename, DEV)
    # OpenAI-style name resolves to the QuickGELU towers
    q = np.asarray(s2i.vectorise("ViT-B/32", ["a photo of a cat"], device=DEV))
    assert q.shape == (1, 512) and _cos_err(q, txt[:1]) > 1e-4
    assert len(s2i.get_available_models()) >= 3
    s2i.eject_model(name, DEV)


KNOB_SWEEP = [("gemm_nh", 3), ("gemm_nh", 4), ("gemm_nh", 1), ("gemm_tail", 1), ("gemm_wd", 3), ("gemm_wd", 6), ("rs_finalize", 1), ("ln_fold", 0), ("ln_fold", 1),
              ("small_m", 0), ("small_m_grouped", 0), ("row_select", 0), ("attn_waves", 8), ("attn_waves", 4), ("ln_prefetch", 0),
              ("gemm_mt", 2), ("gemm_mt", 6), ("gemm_cgroup", 0), ("xcd_band", 0), ("attn_proj", 0), ("attn_proj", 1)]


def test_every_kernel_family_knob_keeps_vectorise_right(s2):
    """every mq_tune key that switches a kernel family, at its NON-default value, through vectorise() itself (registry ViT-B/32 on
    seeded weights; 40 images = 2 000 token rows: the tiled GEMM families; 33 texts; one lone query: the skinny families): still the oracle's
    embeddings inside the north-star tolerance, and within 3e-4 of the default configuration's."""
    s2i, _ = s2
    from marqo_amd import _lib as L
    from marqo_amd.engine import archs, synthetic
    os.environ["MARQO_AMD_SYNTHETIC_WEIGHTS"] = "1"
    lib = L.load()
    name = "open_clip/ViT-B-32/laion2b_s34b_b79k"
    rng = np.random.default_rng(5)
    pil = [Image.fromarray(rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)) for _ in range(40)]
    texts = [f"a photo of object number {i} " + "with many details " * (i % 7) for i in range(33)]
    run = lambda: (np.asarray(s2i.vectorise(name, pil, device=DEV, modality=s2i.Modality.IMAGE)), np.asarray(s2i.vectorise(name, texts, device=DEV)),
                   np.asarray(s2i.vectorise(name, texts[3], device=DEV)))
    base_i, base_t, base_q = run()
    varch, tarch = archs.resolve_open_clip("ViT-B-32")
    sdc = synthetic.random_open_clip_state_dict(vision=varch, text=tarch, seed=0)
    refi = O.vit_forward(sdc, O.VitConfig(224, 32, 768, 12, 12, 3072, 512), torch.from_numpy(np.stack([OP.clip_transform(np.asarray(p)) for p in pil[:6]]))).numpy()
    assert _cos_err(base_i[:6], refi) < COS_TOL and _cos_err(base_q, base_t[3:4]) < 3e-4
    defaults = {k: 0 for k, _ in KNOB_SWEEP}
    defaults.update(gemm_cgroup=8, ln_fold=2, small_m=80, small_m_grouped=320, row_select=1, ln_prefetch=1, xcd_band=1, attn_proj=128)
    worst = {}
    for key, value in KNOB_SWEEP:
        try:
            L.check(lib.mq_tune(key.encode(), value), "mq_tune")
            i, t, q = run()
        finally:
            L.check(lib.mq_tune(key.encode(), defaults[key]), "mq_tune")
        worst[(key, value)] = max(_cos_err(i, base_i), _cos_err(t, base_t), _cos_err(q, base_q))
        assert _cos_err(i[:6], refi) < COS_TOL, (key, value)
        assert worst[(key, value)] < 3e-4, (key, value, worst[(key, value)])
    i, t, q = run()
    assert np.array_equal(i, base_i) and np.array_equal(t, base_t) and np.array_equal(q, base_q)      # the defaults are back
    print("knob sweep, max 1 - cos vs the default configuration:", {f"{k}={v}": f"{e:.1e}" for (k, v), e in worst.items()})


def test_preprocess_slab_views_are_encoded_without_a_gather(s2):
    """round 6: `.preprocess` (called image by image from the reference's download threads, add_docs.py:129-141) returns a
    Tensor (3, 224, 224) on the device — the add_documents contract (test_add_documents_combined.py:411-439) — that is a VIEW of a per-model slab;
    a list of such views is encoded from the slab slices themselves.  Same bits as the gather of the same tensors, for neighbouring slots, for a
    shuffled list (few long runs or the stacked fallback), for copies (no slot information: fallback), from several threads at once."""
    import threading
    s2i, _ = s2
    os.environ["MARQO_AMD_SYNTHETIC_WEIGHTS"] = "1"
    name = "open_clip/ViT-B-32/laion2b_s34b_b79k"
    props = s2i.get_model_properties_from_registry(name)
    model, pre = s2i.load_multimodal_model_and_get_preprocessors(name, props, DEV)
    enc = s2i.get_available_models()[s2i._create_model_cache_key(name, DEV, props)]["model"]
    rng = np.random.default_rng(11)
    pil = [Image.fromarray(rng.integers(0, 256, (200 + 3 * i, 260 - 2 * i, 3), dtype=np.uint8)) for i in range(40)]
    views = [pre["image"](p) for p in pil]
    for v in views:
        assert isinstance(v, torch.Tensor) and tuple(v.shape) == (3, 224, 224) and v.dtype == torch.float32 and v.is_cuda and v.to(DEV) is v
    assert all(v._mq_block is views[0]._mq_block and v._mq_slot == views[0]._mq_slot + k for k, v in enumerate(views))      # side by side in one block
    want = enc.encode_image(torch.stack([v.clone() for v in views]))                       # the batch form (no slot information)
    got = enc.encode_image(views)
    assert np.array_equal(got, want)
    assert enc.image_input_processed.data_ptr() == views[0].data_ptr()                      # the tower read the slab itself
    ref = O.vit_forward(__import__("marqo_amd.engine.synthetic", fromlist=["x"]).random_open_clip_state_dict(
        vision=__import__("marqo_amd.engine.archs", fromlist=["x"]).resolve_open_clip("ViT-B-32")[0],
        text=__import__("marqo_amd.engine.archs", fromlist=["x"]).resolve_open_clip("ViT-B-32")[1], seed=0),
        O.VitConfig(224, 32, 768, 12, 12, 3072, 512), torch.from_numpy(np.stack([OP.clip_transform(np.asarray(p)) for p in pil[:4]]))).numpy()
    assert _cos_err(got[:4], ref) < COS_TOL
    order = list(rng.permutation(40))
    assert np.array_equal(enc.encode_image([views[i] for i in order]), want[order])         # scattered slots: the stacked fallback
    two_runs = views[20:] + views[:20]
    assert np.array_equal(enc.encode_image(two_runs), np.concatenate([want[20:], want[:20]]))
    mixed = [v if i % 2 else v.clone() for i, v in enumerate(views)]
    assert np.array_equal(enc.encode_image(mixed), want)                                    # copies carry no slot: fallback
    assert np.array_equal(np.asarray(s2i.vectorise_ndarray(name, views, model_properties=props, device=DEV, modality=s2i.Modality.IMAGE)), want)
    # four "download threads" preprocess at once (slots interleave between them), each document batch is then encoded: still right
    outs, errs = {}, []

    def worker(t):
        try:
            mine = [pre["image"](pil[i]) for i in range(t, 40, 4)]
            outs[t] = enc.encode_image(mine)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs
    for t in range(4):
        assert np.array_equal(outs[t], want[t::4])
    # an in-place edit of a view IS an edit of the slab: the tower sees it
    views[3].mul_(0.5)
    assert not np.array_equal(enc.encode_image(views)[3], want[3])


def test_hf_xlm_roberta_from_disk(s2, tmp_path):
    """an XLM-RoBERTa checkpoint directory (multilingual-e5 layout: config.json model_type xlm-roberta, `roberta.`-prefixed
    safetensors, sentencepiece.bpe.model) through the `hf` loader: SentencePiece tokeniser on the host, BERT tower with the
    position offset on the GPU, against the fp32 oracle on the same ids"""
    s2i, root = s2
    import sentencepiece as spm
    from safetensors.torch import save_file
    from tests.test_tokenizers import CORPUS, SENTENCES
    d = root / "hf" / "acme" / "tiny-xlmr"
    d.mkdir(parents=True, exist_ok=True)
    corpus = tmp_path / "corpus.txt"
    corpus.write_text("\n".join([" ".join(CORPUS)] * 20 + SENTENCES[:6] * 5), encoding="utf-8")
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(d / "sentencepiece.bpe"), vocab_size=120, model_type="unigram",
                                   character_coverage=1.0, hard_vocab_limit=False, minloglevel=2)
    from marqo_amd.engine.tokenizers import XlmRobertaTokenizer
    tok = XlmRobertaTokenizer(str(d))
    cfg = O.BertConfig(vocab=tok.vocab_size, max_pos=66, width=128, layers=2, heads=2, mlp_dim=256, ln_eps=1e-5, pos_offset=2)
    sd = O.synthetic_bert_state_dict(cfg, seed=4)
    sd["embeddings.token_type_embeddings.weight"] = sd["embeddings.token_type_embeddings.weight"][:1].clone()  # type_vocab_size 1
    save_file({"roberta." + k: v.contiguous() for k, v in sd.items()}, str(d / "model.safetensors"))
    (d / "config.json").write_text(json.dumps({"model_type": "xlm-roberta", "vocab_size": tok.vocab_size, "max_position_embeddings": 66,
                                               "hidden_size": 128, "num_hidden_layers": 2, "num_attention_heads": 2, "intermediate_size": 256,
                                               "hidden_act": "gelu", "layer_norm_eps": 1e-5, "pad_token_id": 1, "type_vocab_size": 1}))
    props = {"name": "acme/tiny-xlmr", "dimensions": 128, "tokens": 32, "type": "hf"}
    texts = ["query: how much protein should a female eat", "naïve café über straße", "東京 photos 2024", "fox"]
    out = s2i.vectorise("tiny-xlmr", texts, model_properties=props, device=DEV)
    t = tok(texts, max_length=32)
    ref = O.hf_encode(sd, cfg, torch.from_numpy(t["input_ids"]), torch.from_numpy(t["attention_mask"])).numpy()
    assert np.asarray(out).shape == (4, 128) and _cos_err(out, ref) < COS_TOL
    model = s2i.get_available_models()[s2i._create_model_cache_key("tiny-xlmr", DEV, props)]["model"]
    assert type(model._tokenizer).__name__ == "XlmRobertaTokenizer" and model.arch.pos_offset == 2
    assert type(model._device_tokenizer).__name__ == "DeviceSentencePieceTokenizer"   # the unigram Viterbi runs on the GPU (K14)


def test_mpnet_from_disk(s2, tmp_path):
    """An MPNet checkpoint (the hf/all-mpnet-base-* registry family) through the 'hf' loader: config.json model_type 'mpnet', MPNetModel
    tensor names under the `mpnet.` task prefix, vocab.txt with <s> / <pad> / </s> specials -> WordPiece between <s> and </s> (on the
    device: K14), positions from 2, the relative-position bias inside the attention kernel — against the fp32 oracle on the ids
    transformers' own MPNetTokenizer produces"""
    s2i, root = s2
    from safetensors.torch import save_file
    from transformers import MPNetTokenizer
    from tests.test_tokenizers import _bert_vocab
    d = root / "hf" / "acme" / "tiny-mpnet"
    d.mkdir(parents=True, exist_ok=True)
    toks = ["<s>", "<pad>", "</s>", "<unk>"] + [t for t in _bert_vocab() if t not in ("[PAD]", "[CLS]", "[SEP]", "[MASK]")] + ["<mask>"]
    (d / "vocab.txt").write_text("\n".join(toks) + "\n", encoding="utf-8")
    (d / "tokenizer_config.json").write_text(json.dumps({"do_lower_case": True, "cls_token": "<s>", "sep_token": "</s>", "pad_token": "<pad>",
                                                         "unk_token": "[UNK]", "mask_token": "<mask>"}))
    cfg = O.BertConfig(vocab=len(toks), max_pos=64, width=128, layers=2, heads=2, mlp_dim=256, ln_eps=1e-5, pos_offset=2)
    sd = O.synthetic_mpnet_state_dict(cfg, seed=8)
    save_file({"mpnet." + k: v.contiguous() for k, v in sd.items()}, str(d / "model.safetensors"))
    (d / "config.json").write_text(json.dumps({"model_type": "mpnet", "vocab_size": len(toks), "max_position_embeddings": 66, "hidden_size": 128,
                                               "num_hidden_layers": 2, "num_attention_heads": 2, "intermediate_size": 256, "hidden_act": "gelu",
                                               "layer_norm_eps": 1e-5, "pad_token_id": 1, "relative_attention_num_buckets": 32}))
    props = {"name": "acme/tiny-mpnet", "dimensions": 128, "tokens": 32, "type": "hf"}
    texts = ["query: how much protein should a female eat", "the quick brown fox jumps over the lazy dog", "naive cafe uber", "fox",
             "the fox </s> the dog"]
    out = s2i.vectorise("tiny-mpnet", texts, model_properties=props, device=DEV)
    hf = MPNetTokenizer(str(d / "vocab.txt"), do_lower_case=True)
    t = hf(texts, padding=True, truncation=True, max_length=32, return_tensors="pt")
    ref = O.hf_encode(sd, cfg, t["input_ids"], t["attention_mask"]).numpy()
    assert np.asarray(out).shape == (5, 128) and _cos_err(out, ref) < COS_TOL
    model = s2i.get_available_models()[s2i._create_model_cache_key("tiny-mpnet", DEV, props)]["model"]
    assert model.arch.rel_buckets == 32 and model.arch.pos_offset == 2 and model._tokenizer.cls == "<s>"
    assert type(model._device_tokenizer).__name__ == "DeviceWordPieceTokenizer"
    one = s2i.vectorise("tiny-mpnet", texts[1], model_properties=props, device=DEV)       # the single-query route
    assert _cos_err(one, ref[1:2]) < COS_TOL


def test_multilingual_clip_with_hf_text_tower_from_disk(s2, tmp_path, monkeypatch):
    """open_clip CustomTextCLIP with an XLM-RoBERTa text tower (the open_clip/xlm-roberta-*-ViT-* registry family) through the 'open_clip'
    loader: `visual.*` + `text.transformer.*` + `text.proj.*` tensors, the SentencePiece model next to the checkpoint, open_clip's
    HFTokenizer semantics (clean, pad to ctx with <pad>), unigram Viterbi on the device — text and image against the fp32 oracle"""
    s2i, root = s2
    import sentencepiece as spm
    from safetensors.torch import save_file
    from marqo_amd.engine import archs as A
    from marqo_amd.engine.tokenizers import XlmRobertaTokenizer
    from tests.test_tokenizers import CORPUS, SENTENCES
    S, P, W, Lyr, H, Fd, D, ctx = 64, 16, 128, 2, 2, 256, 64, 32
    d = tmp_path / "tiny-xlmr-clip"
    d.mkdir()
    (d / "corpus.txt").write_text("\n".join([" ".join(CORPUS)] * 20 + SENTENCES[:6] * 5), encoding="utf-8")
    spm.SentencePieceTrainer.train(input=str(d / "corpus.txt"), model_prefix=str(d / "sentencepiece.bpe"), vocab_size=120, model_type="unigram",
                                   character_coverage=1.0, hard_vocab_limit=False, minloglevel=2)
    tok = XlmRobertaTokenizer(str(d))
    bert = A.BertArch(vocab=tok.vocab_size, max_pos=64, width=W, layers=Lyr, heads=H, mlp_dim=Fd, ln_eps=1e-5, pos_offset=2, type_vocab=1)
    tarch = A.HfClipTextArch(bert=bert, out_dim=D, ctx=ctx)
    vcfg = O.VitConfig(S, P, W, Lyr, H, Fd, D)
    bcfg = O.BertConfig(vocab=tok.vocab_size, max_pos=66, width=W, layers=Lyr, heads=H, mlp_dim=Fd, ln_eps=1e-5, pos_offset=2)
    sd = O.synthetic_vit_state_dict(vcfg, seed=3)
    enc = O.synthetic_bert_state_dict(bcfg, seed=4)
    enc["embeddings.token_type_embeddings.weight"] = enc["embeddings.token_type_embeddings.weight"][:1].clone()
    sd.update({"text.transformer." + k: v for k, v in enc.items()})
    g = torch.Generator().manual_seed(5)
    sd["text.proj.0.weight"] = torch.randn(tarch.proj_hidden, W, generator=g) / W ** 0.5
    sd["text.proj.2.weight"] = torch.randn(D, tarch.proj_hidden, generator=g) / tarch.proj_hidden ** 0.5
    save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "open_clip_model.safetensors"))
    monkeypatch.setitem(A.OPEN_CLIP_ARCHS, "tiny-xlmr-ViT", (A.VitArch(S, P, W, Lyr, H, Fd, D), tarch))
    props = {"name": "tiny-xlmr-ViT", "dimensions": D, "type": "open_clip", "localpath": str(d / "open_clip_model.safetensors")}
    texts = ["A photo of a CAT!", "naïve café über straße", "東京 photos 2024", "fox " * 60, "Tom &amp; Jerry   together"]
    out = np.asarray(s2i.vectorise("tiny-xlmr-clip", texts, model_properties=props, device=DEV))
    from marqo_amd.engine.tokenizers import _clean_text
    t = tok([_clean_text(x) for x in texts], max_length=ctx)
    ids = np.full((len(texts), ctx), 1, dtype=np.int64)
    ids[:, :t["input_ids"].shape[1]] = t["input_ids"]
    ref = O.hf_clip_text_forward(sd, bcfg, torch.from_numpy(ids)).numpy()
    assert out.shape == (5, D) and _cos_err(out, ref) < COS_TOL
    model = s2i.get_available_models()[s2i._create_model_cache_key("tiny-xlmr-clip", DEV, props)]["model"]
    assert type(model.text).__name__ == "HfClipTextTower" and type(model.tokenizer).__name__ == "HfClipTokenizer"
    assert type(model._device_tokenizer).__name__ == "DeviceSentencePieceTokenizer"
    assert np.array_equal(model.tokenizer(texts), ids)                      # open_clip HFTokenizer output: ids padded to ctx with <pad>
    one = np.asarray(s2i.vectorise("tiny-xlmr-clip", texts[0], model_properties=props, device=DEV))
    assert _cos_err(one, ref[:1]) < COS_TOL
    rng = np.random.default_rng(2)
    pil = [Image.fromarray(rng.integers(0, 256, (80, 100, 3), dtype=np.uint8)) for _ in range(3)]
    img = np.asarray(s2i.vectorise("tiny-xlmr-clip", pil, model_properties=props, device=DEV, modality=s2i.Modality.IMAGE))
    refi = O.vit_forward(sd, vcfg, torch.from_numpy(np.stack([OP.clip_transform(np.asarray(p), S) for p in pil]))).numpy()
    assert _cos_err(img, refi) < COS_TOL


def test_clip_with_roberta_text_tower_from_disk(s2, tmp_path, monkeypatch):
    """open_clip/roberta-ViT-B-32 family: the same HF text tower with RoBERTa's byte-level BPE (vocab.json + merges.txt next to the
    checkpoint, host tokeniser) and a QuickGELU vision tower"""
    s2i, root = s2
    from safetensors.torch import save_file
    from marqo_amd.engine import archs as A
    from marqo_amd.engine.tokenizers import RobertaBpeTokenizer, _clean_text
    from tests.test_tokenizers import CORPUS, SENTENCES, _train_byte_level_bpe
    S, P, W, Lyr, H, Fd, D, ctx = 64, 16, 128, 2, 2, 256, 64, 32
    d = tmp_path / "tiny-roberta-clip"
    d.mkdir()
    vocab, merges = _train_byte_level_bpe(CORPUS + SENTENCES, 120)
    (d / "vocab.json").write_text(json.dumps(vocab), encoding="utf-8")
    (d / "merges.txt").write_text("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n", encoding="utf-8")
    tok = RobertaBpeTokenizer(str(d))
    bert = A.BertArch(vocab=50265, max_pos=64, width=W, layers=Lyr, heads=H, mlp_dim=Fd, ln_eps=1e-5, pos_offset=2, type_vocab=1)
    tarch = A.HfClipTextArch(bert=bert, out_dim=D, ctx=ctx)
    vcfg = O.VitConfig(S, P, W, Lyr, H, Fd, D, quick_gelu=True)
    bcfg = O.BertConfig(vocab=50265, max_pos=66, width=W, layers=Lyr, heads=H, mlp_dim=Fd, ln_eps=1e-5, pos_offset=2)
    sd = O.synthetic_vit_state_dict(vcfg, seed=3)
    enc = O.synthetic_bert_state_dict(bcfg, seed=4)
    enc["embeddings.token_type_embeddings.weight"] = enc["embeddings.token_type_embeddings.weight"][:1].clone()
    sd.update({"text.transformer." + k: v for k, v in enc.items()})
    g = torch.Generator().manual_seed(5)
    sd["text.proj.0.weight"] = torch.randn(tarch.proj_hidden, W, generator=g) / W ** 0.5
    sd["text.proj.2.weight"] = torch.randn(D, tarch.proj_hidden, generator=g) / tarch.proj_hidden ** 0.5
    save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "open_clip_model.safetensors"))
    monkeypatch.setitem(A.OPEN_CLIP_ARCHS, "tiny-roberta-ViT", (A.VitArch(S, P, W, Lyr, H, Fd, D, quick_gelu=True), tarch))
    props = {"name": "tiny-roberta-ViT", "dimensions": D, "type": "open_clip", "localpath": str(d / "open_clip_model.safetensors")}
    texts = ["A photo of a CAT!", "it's the quick brown fox", "naïve café", "fox " * 60]
    out = np.asarray(s2i.vectorise("tiny-roberta-clip", texts, model_properties=props, device=DEV))
    t = tok([_clean_text(x) for x in texts], max_length=ctx)
    ids = np.full((len(texts), ctx), 1, dtype=np.int64)
    ids[:, :t["input_ids"].shape[1]] = t["input_ids"]
    assert out.shape == (4, D) and _cos_err(out, O.hf_clip_text_forward(sd, bcfg, torch.from_numpy(ids)).numpy()) < COS_TOL
    model = s2i.get_available_models()[s2i._create_model_cache_key("tiny-roberta-clip", DEV, props)]["model"]
    assert type(model.tokenizer.hf).__name__ == "RobertaBpeTokenizer" and model._device_tokenizer is None and model.vision_arch.quick_gelu
    rng = np.random.default_rng(2)
    pil = [Image.fromarray(rng.integers(0, 256, (80, 100, 3), dtype=np.uint8)) for _ in range(2)]
    img = np.asarray(s2i.vectorise("tiny-roberta-clip", pil, model_properties=props, device=DEV, modality=s2i.Modality.IMAGE))
    refi = O.vit_forward(sd, vcfg, torch.from_numpy(np.stack([OP.clip_transform(np.asarray(p), S) for p in pil]))).numpy()
    assert _cos_err(img, refi) < COS_TOL


def test_multilingual_clip_loader_type(s2, tmp_path, monkeypatch):
    """the reference's `multilingual_clip` loader type (clip_utils.py:521-597): an OpenAI CLIP image tower paired with an M-CLIP text
    encoder (HF encoder -> masked mean -> one biased Linear) from its own checkpoint directory; registry-name call through vectorise()"""
    s2i, root = s2
    import sentencepiece as spm
    from safetensors.torch import save_file
    from marqo_amd.engine import archs as A
    from marqo_amd.engine.tokenizers import XlmRobertaTokenizer
    from marqo_amd.s2_inference import open_clip_model as M
    from tests.test_tokenizers import CORPUS, SENTENCES
    monkeypatch.setenv("MARQO_AMD_SYNTHETIC_WEIGHTS", "1")      # the image tower: random-init ViT-B/32 (QuickGELU, OpenAI naming)
    W, Lyr, H, Fd, D = 128, 2, 2, 256, 512
    d = root / "hf" / "M-CLIP" / "XLM-Roberta-Large-Vit-B-32"
    d.mkdir(parents=True, exist_ok=True)
    (tmp_path / "corpus.txt").write_text("\n".join([" ".join(CORPUS)] * 20 + SENTENCES[:6] * 5), encoding="utf-8")
    spm.SentencePieceTrainer.train(input=str(tmp_path / "corpus.txt"), model_prefix=str(d / "sentencepiece.bpe"), vocab_size=120,
                                   model_type="unigram", character_coverage=1.0, hard_vocab_limit=False, minloglevel=2)
    tok = XlmRobertaTokenizer(str(d))
    small = A.BertArch(vocab=tok.vocab_size, max_pos=64, width=W, layers=Lyr, heads=H, mlp_dim=Fd, ln_eps=1e-5, pos_offset=2, type_vocab=1)
    monkeypatch.setitem(M._MCLIP_BASES, "xlm-roberta-large", small)
    bcfg = O.BertConfig(vocab=tok.vocab_size, max_pos=66, width=W, layers=Lyr, heads=H, mlp_dim=Fd, ln_eps=1e-5, pos_offset=2)
    enc = O.synthetic_bert_state_dict(bcfg, seed=9)
    enc["embeddings.token_type_embeddings.weight"] = enc["embeddings.token_type_embeddings.weight"][:1].clone()
    sd = {"transformer." + k: v for k, v in enc.items()}
    g = torch.Generator().manual_seed(10)
    sd["LinearTransformation.weight"] = torch.randn(D, W, generator=g) / W ** 0.5
    sd["LinearTransformation.bias"] = 0.1 * torch.randn(D, generator=g)
    save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "model.safetensors"))
    (d / "config.json").write_text(json.dumps({"modelBase": "xlm-roberta-large", "transformerDimensions": W, "numDims": D}))
    name = "multilingual-clip/XLM-Roberta-Large-Vit-B-32"
    texts = ["A photo of a CAT!", "naïve café über straße", "東京 photos 2024", "fox " * 80]
    out = np.asarray(s2i.vectorise(name, texts, device=DEV))
    t = tok(texts, max_length=64)
    ref = O.mclip_text_forward(sd, bcfg, torch.from_numpy(t["input_ids"]), torch.from_numpy(t["attention_mask"])).numpy()
    assert out.shape == (4, D) and _cos_err(out, ref) < COS_TOL
    model = s2i.get_available_models()[s2i._create_model_cache_key(name, DEV, s2i.get_model_properties_from_registry(name))]["model"]
    assert type(model).__name__ == "MULTILINGUAL_CLIP" and type(model.text).__name__ == "MclipTextTower" and model.vision_arch.quick_gelu
    assert type(model._device_tokenizer).__name__ == "DeviceSentencePieceTokenizer"
    raw = np.asarray(s2i.vectorise(name, texts[:2], device=DEV, normalize_embeddings=False))
    assert _cos_err(raw, O.mclip_text_forward(sd, bcfg, torch.from_numpy(t["input_ids"][:2]), torch.from_numpy(t["attention_mask"][:2]),
                                              normalize=False).numpy()) < COS_TOL and not np.allclose(np.linalg.norm(raw, axis=1), 1, atol=1e-3)
    rng = np.random.default_rng(2)
    pil = [Image.fromarray(rng.integers(0, 256, (240, 300, 3), dtype=np.uint8)) for _ in range(2)]
    img = np.asarray(s2i.vectorise(name, pil, device=DEV, modality=s2i.Modality.IMAGE))
    assert img.shape == (2, D) and np.allclose(np.linalg.norm(img, axis=1), 1, atol=1e-5)
    from marqo_amd.s2_inference.errors import InternalError
    with pytest.raises(InternalError):
        M.MULTILINGUAL_CLIP(name, device=None)
    s2i.eject_model(name, DEV)


def test_onnx_loader_types_are_served_by_the_same_towers(s2, monkeypatch):
    """`clip_onnx` / `sbert_onnx` registry names are ONNX exports of checkpoints the engine already runs: same embeddings as the entry they
    were exported from (the reference's constructors and return types kept)"""
    s2i, root = s2
    monkeypatch.setenv("MARQO_AMD_SYNTHETIC_WEIGHTS", "1")
    texts = ["a photo of a cat", "a dog on the beach"]
    for onnx, plain in (("onnx32/open_clip/ViT-B-32/laion400m_e32", "open_clip/ViT-B-32/laion400m_e32"), ("onnx16/openai/ViT-L/14", "ViT-L/14")):
        a = np.asarray(s2i.vectorise(onnx, texts, device=DEV))
        b = np.asarray(s2i.vectorise(plain, texts, device=DEV))
        assert a.shape == b.shape and np.array_equal(a, b), onnx
        key = s2i._create_model_cache_key(onnx, DEV, s2i.get_model_properties_from_registry(onnx))
        assert type(s2i.get_available_models()[key]["model"]).__name__ == "CLIP_ONNX"
        s2i.eject_model(onnx, DEV), s2i.eject_model(plain, DEV)
    a = np.asarray(s2i.vectorise("onnx/all-MiniLM-L6-v2", texts, device=DEV))
    b = np.asarray(s2i.vectorise("sentence-transformers/all-MiniLM-L6-v2", texts, device=DEV))
    assert a.shape == (2, 384) and np.array_equal(a, b)
    from marqo_amd.s2_inference.sbert_utils import SBERT_ONNX
    m = SBERT_ONNX("sentence-transformers/all-MiniLM-L6-v2", device=DEV, embedding_dim=384, max_seq_length=256)
    out = m.encode(texts)
    assert isinstance(out, torch.Tensor) and out.device.type == "cpu" and np.allclose(out.numpy(), a, atol=1e-6)
    from marqo_amd.s2_inference.errors import InvalidModelPropertiesError
    from marqo_amd.s2_inference.open_clip_model import CLIP_ONNX
    with pytest.raises(InvalidModelPropertiesError):
        CLIP_ONNX("onnx32/open_clip/RN50/openai", device=DEV, embedding_dim=1024).load()


def test_clipa_family_through_the_loader(s2, tmp_path, monkeypatch):
    """a CLIPA-style open_clip entry end to end: avg-pool / no-ln_pre vision tower behind the bilinear-squash preprocessing with ImageNet
    statistics (the architecture's own preprocess config), and open_clip's HFTokenizer(bert-base-uncased, strip_sep_token=True) — WordPiece
    ids, [SEP] replaced by 0, padded to the context — feeding the unmasked last-position text tower"""
    s2i, root = s2
    from safetensors.torch import save_file
    from marqo_amd.engine import archs as A
    from marqo_amd.engine.tokenizers import WordPieceTokenizer, _clean_text
    from tests.test_tokenizers import _bert_vocab
    S, P, W, Lyr, H, Fd, D, ctx = 64, 16, 128, 2, 2, 256, 64, 16
    d = tmp_path / "tiny-clipa"
    d.mkdir()
    vocab = _bert_vocab()
    (d / "vocab.txt").write_text("\n".join(sorted(vocab, key=vocab.get)) + "\n")
    vcfg = O.VitConfig(S, P, W, Lyr, H, Fd, D, ln_pre=False, pool="avg")
    tcfg = O.ClipTextConfig(vocab=len(vocab), ctx=ctx, width=W, layers=Lyr, heads=H, mlp_dim=Fd, out_dim=D, causal=False)
    sd = {k: v for k, v in O.synthetic_vit_state_dict(vcfg, seed=31).items() if not k.startswith("visual.ln_pre.")}
    sd.update(O.synthetic_clip_text_state_dict(tcfg, seed=32))
    save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "open_clip_model.safetensors"))
    monkeypatch.setitem(A.OPEN_CLIP_ARCHS, "tiny-CLIPA", (
        A.VitArch(S, P, W, Lyr, H, Fd, D, pool="avg", ln_pre=False, preprocessor="CLIPA"),
        A.ClipTextArch(vocab=len(vocab), ctx=ctx, width=W, layers=Lyr, heads=H, mlp_dim=Fd, out_dim=D, causal=False,
                       hf_tokenizer="bert-base-uncased", strip_sep=True)))
    props = {"name": "tiny-CLIPA", "dimensions": D, "type": "open_clip", "localpath": str(d / "open_clip_model.safetensors"),
             "image_preprocessor": "CLIPA"}
    texts = ["A photo of a CAT!", "the quick brown fox jumps over the lazy dog and the cat " * 2, "Tom &amp; Jerry", "fox"]
    out = np.asarray(s2i.vectorise("tiny-clipa", texts, model_properties=props, device=DEV))
    wp = WordPieceTokenizer(vocab)
    ids = np.zeros((len(texts), ctx), dtype=np.int64)
    for i, t in enumerate(texts):
        e = wp.encode(_clean_text(t), max_length=ctx)
        ids[i, :len(e)] = e
    ids[ids == wp.sep_id] = 0
    assert (ids[:, 0] == wp.cls_id).all() and not (ids == wp.sep_id).any()
    model = s2i.get_available_models()[s2i._create_model_cache_key("tiny-clipa", DEV, props)]["model"]
    assert np.array_equal(model.tokenizer(texts), ids)
    assert out.shape == (4, D) and _cos_err(out, O.clip_text_forward(sd, tcfg, torch.from_numpy(ids)).numpy()) < COS_TOL
    assert model.preprocess_config["interpolation"] == "bilinear" and model.preprocess_config["resize_mode"] == "squash"
    rng = np.random.default_rng(3)
    pil = [Image.fromarray(rng.integers(0, 256, (90, 130, 3), dtype=np.uint8)) for _ in range(3)]
    img = np.asarray(s2i.vectorise("tiny-clipa", pil, model_properties=props, device=DEV, modality=s2i.Modality.IMAGE))
    px = np.stack([OP.to_tensor_normalize(OP.squash_pil_image(p, S, S, bilinear=True), (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)) for p in pil])
    assert _cos_err(img, O.vit_forward(sd, vcfg, torch.from_numpy(px)).numpy()) < COS_TOL


def test_siglip_from_disk_text_and_image(s2, tmp_path, monkeypatch):
    """A SigLIP checkpoint through the loader: open_clip / timm tensor names (visual.trunk.*, text.*), SentencePiece tokenizer with
    canonicalize, SigLIP preprocessing (squash to S x S, mean = std = 0.5), 'open_clip' loader type — against the fp32 oracle.
    (Real SigLIP shapes come from the architecture table; a small one is registered for this test.)"""
    s2i, root = s2
    import sentencepiece as spm
    from safetensors.torch import save_file
    from marqo_amd.engine import archs as A
    from marqo_amd.engine.tokenizers import SiglipTokenizer
    from tests.test_tokenizers import CORPUS, SENTENCES
    S, P, W, Lyr, H, Fd, ctx = 64, 16, 128, 2, 2, 256, 64
    d = tmp_path / "tiny-siglip"
    d.mkdir()
    (d / "corpus.txt").write_text("\n".join([" ".join(CORPUS)] * 20 + SENTENCES[:6] * 5).lower(), encoding="utf-8")
    spm.SentencePieceTrainer.train(input=str(d / "corpus.txt"), model_prefix=str(d / "spiece"), vocab_size=120, model_type="unigram",
                                   character_coverage=1.0, hard_vocab_limit=False, minloglevel=2, pad_id=0, eos_id=1, unk_id=2, bos_id=-1)
    tok = SiglipTokenizer(str(d / "spiece.model"), context_length=ctx)
    V = tok.vocab_size
    vcfg = O.SiglipVitConfig(S, P, W, Lyr, H, Fd)
    tcfg = O.SiglipTextConfig(V, ctx, W, Lyr, H, Fd, W)
    sd = O.synthetic_siglip_state_dict(vcfg, tcfg, seed=6)
    save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "open_clip_model.safetensors"))
    monkeypatch.setitem(A.OPEN_CLIP_ARCHS, "tiny-SigLIP", (
        A.VitArch(S, P, W, Lyr, H, Fd, W, ln_eps=1e-6, pool="map"),
        A.ClipTextArch(vocab=V, ctx=ctx, width=W, layers=Lyr, heads=H, mlp_dim=Fd, out_dim=W, ln_eps=1e-6, causal=False,
                       proj_bias=True, prefix="text.", pad_id=1)))
    props = {"name": "tiny-SigLIP", "dimensions": W, "type": "open_clip", "localpath": str(d / "open_clip_model.safetensors"),
             "image_preprocessor": "SigLIP"}
    texts = ["A photo of a CAT!", "the quick_brown fox, jumps over the lazy dog", "marqo is a tensor search engine", "fox " * 100]
    out = s2i.vectorise("tiny-siglip", texts, model_properties=props, device=DEV)
    ref = O.siglip_text_forward(sd, tcfg, torch.from_numpy(tok(texts))).numpy()
    assert np.asarray(out).shape == (4, W) and _cos_err(out, ref) < COS_TOL
    assert np.allclose(np.linalg.norm(np.asarray(out), axis=1), 1.0, atol=1e-5)
    # canonicalize: the same text after lower-casing / punctuation stripping (a one-query call runs the skinny GEMM family, the batch the
    # tiled one: same products, another bf16 summation order)
    assert _cos_err(s2i.vectorise("tiny-siglip", "a photo of a cat", model_properties=props, device=DEV), out[:1]) < 3e-5
    model = s2i.get_available_models()[s2i._create_model_cache_key("tiny-siglip", DEV, props)]["model"]
    assert model.preprocess_config["resize_mode"] == "squash" and model.preprocess_config["mean"] == (0.5, 0.5, 0.5)
    # images: squash = PIL resize((S, S), BICUBIC), no crop
    rng = np.random.default_rng(1)
    pil = [Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for h, w in [(64, 64), (100, 80), (70, 200)]]
    emb = s2i.vectorise("tiny-siglip", pil, model_properties=props, device=DEV, modality=s2i.Modality.IMAGE)
    half = (0.5, 0.5, 0.5)
    px = torch.from_numpy(np.stack([OP.to_tensor_normalize(np.asarray(p.resize((S, S), Image.BICUBIC)), half, half) for p in pil]))
    refi = O.siglip_vit_forward(sd, vcfg, px).numpy()
    assert _cos_err(emb, refi) < COS_TOL
    t = model.preprocess(pil[2])
    assert tuple(t.shape) == (3, S, S) and float((t.cpu() - px[2]).abs().max()) < 1e-6
    # chunking with the squash pipeline: chunk 0 is the whole image squashed, the grid crops are square
    ce, boxes = model.encode_image_chunks([pil[2]], 3, 3, False)
    patches, _ = OP.chunk_image_simple(np.asarray(pil[2]), 3, 3, False)
    pc = [OP.to_tensor_normalize(np.asarray(Image.fromarray(p).resize((S, S), Image.BICUBIC)), half, half) for p in patches]
    refc = O.siglip_vit_forward(sd, vcfg, torch.from_numpy(np.stack(pc))).numpy()
    assert ce.shape == (1, 10, W) and _cos_err(ce[0], refc) < COS_TOL
    # registry / hf-hub names pick the SigLIP pipeline by themselves
    os.environ["MARQO_AMD_SYNTHETIC_WEIGHTS"] = "1"
    try:
        for name in ("open_clip/ViT-B-16-SigLIP/webli", "Marqo/marqo-fashionSigLIP"):
            v = s2i.vectorise(name, ["a red dress", "blue denim jeans"], device=DEV)
            assert np.asarray(v).shape == (2, 768)
            m = next(x["model"] for k, x in s2i.get_available_models().items() if k.startswith(name))
            assert m.preprocess_config["resize_mode"] == "squash" and m.vision_arch.pool == "map" and not m.text_arch.causal
            iv = s2i.vectorise(name, pil[:2], device=DEV, modality=s2i.Modality.IMAGE)
            assert np.asarray(iv).shape == (2, 768) and np.allclose(np.linalg.norm(np.asarray(iv), axis=1), 1.0, atol=1e-5)
            s2i.eject_model(name, DEV)
    finally:
        os.environ.pop("MARQO_AMD_SYNTHETIC_WEIGHTS", None)


@pytest.mark.parametrize("streams", [1, 2, 3])
def test_large_image_call_is_pipelined_in_stages(s2, monkeypatch, streams):
    """>= PIPELINE_MIN images go through host-pack / GPU-encode in stages of about PIPELINE_CHUNK images, alternating between PIPELINE_STREAMS
    HIP streams: same embeddings as one pass, in order — and the SAME BITS whatever the number of streams; mixed PIL / array inputs;
    `image_input_processed` still holds the whole preprocessed batch."""
    s2i, root = s2
    props, sd, vcfg, _ = _tiny_clip(root)
    from marqo_amd.s2_inference import open_clip_model as M
    rng = np.random.default_rng(5)
    imgs = [rng.integers(0, 256, (40 + (i % 7) * 9, 50 + (i % 5) * 11, 3), dtype=np.uint8) for i in range(37)]
    imgs = [Image.fromarray(a) if i % 2 else a for i, a in enumerate(imgs)]
    monkeypatch.setattr(M, "PIPELINE_MIN", 1000)
    whole = np.asarray(s2i.vectorise("tiny-clip", imgs, model_properties=props, device=DEV, modality=s2i.Modality.IMAGE))
    monkeypatch.setattr(M, "PIPELINE_MIN", 16)
    monkeypatch.setattr(M, "PIPELINE_CHUNK", 8)
    assert [b - a for a, b in M._pipeline_stages(37)] == [8, 8, 8, 8, 5]
    model = s2i.get_available_models()[s2i._create_model_cache_key("tiny-clip", DEV, props)]["model"]
    monkeypatch.setattr(M, "PIPELINE_STREAMS", 1)
    one_stream = model.encode_image(imgs)
    monkeypatch.setattr(M, "PIPELINE_STREAMS", streams)
    for helper in (True, False, True):      # (repeated: the side streams' buffers are recycled between calls; towers enqueued by the helper thread or not)
        monkeypatch.setattr(M, "PIPELINE_THREAD", helper)
        monkeypatch.setattr(M, "PIPELINE_CHUNK", 9 if helper else 8)      # (the helper serves stages SMALLER than PIPELINE_CHUNK)
        assert [b - a for a, b in M._pipeline_stages(37)] == ([10, 10, 10, 7] if helper else [8, 8, 8, 8, 5])
        staged = model.encode_image(imgs)
        assert _cos_err(staged, one_stream) < 3e-5 and (helper or np.array_equal(staged, one_stream))
    bad = list(imgs)
    bad[30] = "not an image, not a path"      # a stage that raises on the calling thread while earlier towers sit on the helper: the error comes out, nothing hangs
    with pytest.raises(Exception):
        model.encode_image(bad)
    monkeypatch.setattr(M, "PIPELINE_CHUNK", 8)
    monkeypatch.setattr(M, "PIPELINE_THREAD", False)
    # a call that finds another image call in flight on the model runs in one batch (the other call keeps the GPU busy)
    seen = []
    real_stages = M._pipeline_stages
    monkeypatch.setattr(M, "_pipeline_stages", lambda n_: seen.append(n_) or real_stages(n_))
    model._image_calls += 1
    try:
        assert _cos_err(model.encode_image(imgs), whole) < 3e-5 and not seen
    finally:
        model._image_calls -= 1
    assert model._image_calls == 0
    staged = model.encode_image(imgs)
    assert seen == [37]
    assert staged.shape == whole.shape and _cos_err(staged, whole) < 3e-5      # (8-image stages of 17 tokens cross GEMM kernel families)
    assert tuple(model.image_input_processed.shape) == (37, 64, 64, 3)
    seen.clear()
    dev_rows = model.encode_image(imgs, return_device=True)                   # the ingest path's form: rows stay in HBM, the call stays whole
    assert not seen and _cos_err(dev_rows.cpu().numpy(), whole) < 3e-5
    monkeypatch.setattr(M, "PIPELINE_ALWAYS", True)                           # forced: staged, rows ordered behind the caller's stream
    assert np.array_equal(model.encode_image(imgs, return_device=True).cpu().numpy(), one_stream) and seen == [37]


def test_image_staging_is_bounded_by_bytes(s2, monkeypatch):
    """MARQO_AMD_IMAGE_STAGE_BYTES bounds the pinned / HBM staging of decoded pixels — a call whose images exceed the budget is
    resized in several groups (same pixels, same order, image modes mixed), not packed in one buffer"""
    s2i, root = s2
    props, sd, vcfg, _ = _tiny_clip(root)
    from marqo_amd.engine import preprocess as P
    from marqo_amd.s2_inference import open_clip_model as M
    rng = np.random.default_rng(6)
    arrs = [rng.integers(0, 256, (90 + (i % 4) * 30, 120 + (i % 3) * 25, 4), dtype=np.uint8) for i in range(21)]
    imgs = [Image.fromarray(a, "RGBA") if i % 3 == 0 else Image.fromarray(np.ascontiguousarray(a[..., :3])) for i, a in enumerate(arrs)]
    s2i.vectorise("tiny-clip", imgs[:2], model_properties=props, device=DEV, modality=s2i.Modality.IMAGE)
    model = s2i.get_available_models()[s2i._create_model_cache_key("tiny-clip", DEV, props)]["model"]
    whole = model.encode_image(imgs)
    whole_px = model.image_input_processed.clone()
    calls = []
    real = P.PackedImages.__init__

    def spy(self, images, device, channels=3, **kw):
        calls.append(len(images))
        real(self, images, device, channels, **kw)
    monkeypatch.setattr(P.PackedImages, "__init__", spy)
    monkeypatch.setattr(M, "STAGE_BYTES", 200_000)            # ~3 images per group
    split = model.encode_image(imgs)
    assert len(calls) >= 5 and sum(calls) == 21 and max(calls) <= 6
    assert torch.equal(model.image_input_processed, whole_px) and _cos_err(split, whole) < 3e-5


