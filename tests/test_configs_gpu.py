"""BASELINE.json configs 4 and 5 in miniature, end to end on the GPU against the CPU oracle.

config 4 — add_documents bulk ingest of mixed text+image documents: BulkVectoriser (one vectorise call per modality, device
           tokenisation + device preprocessing + towers), multimodal-combination of the two sub-embeddings on the device
           (mq_weighted_combine), results in document order.  (The 8-GPU sharding of the same flush is covered on CPU by
           tests/test_ingest.py::test_sharded_flush_world_size_2.)
config 5 — fp8 (e4m3 MX-MFMA) towers + on-GPU image chunking (patch localisation): every image becomes 1 + hn*wn crops that go
           through the fp8 ViT; checked against the oracle chunker + fp32 oracle tower with the honest fp8 bound."""
import numpy as np
import pytest
import torch
from PIL import Image

from oracle import combine as OC
from oracle import preprocess as OP
from oracle import towers as O
from tests.test_s2_inference_gpu import DEV, _cos_err, _tiny_clip, s2  # noqa: F401  (s2 is a fixture)

pytestmark = pytest.mark.gpu


def _docs(n, seed):
    rng = np.random.default_rng(seed)
    words = "a photo of the quick brown fox dog cat marqo tensor search engine image text lazy jumps over".split()
    docs = []
    for i in range(n):
        h, w = int(rng.integers(40, 120)), int(rng.integers(40, 120))
        docs.append({"_id": f"doc{i}", "text": " ".join(rng.choice(words, size=int(rng.integers(2, 12)))),
                     "image": Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))})
    return docs


def test_config4_bulk_ingest_mixed_documents(s2):
    s2i, root = s2
    props, sd, vcfg, tcfg = _tiny_clip(root)
    from marqo_amd import combine as MC
    from marqo_amd.ingest import BulkVectoriser
    docs = _docs(37, seed=11)
    bv = BulkVectoriser("tiny-clip", DEV, model_properties=props, max_pending=50)  # auto-flushes once mid-way
    for d in docs:
        bv.add((d["_id"], "text"), d["text"], s2i.Modality.TEXT)
        bv.add((d["_id"], "image"), d["image"], s2i.Modality.IMAGE)
    emb = bv.flush()
    assert len(emb) == 2 * len(docs) and bv.pending() == 0
    weights = {"text": 0.3, "image": 0.7}
    got = MC.combine_multimodal_fields([{k: emb[(d["_id"], k)] for k in weights} for d in docs], weights, normalize=True, device=DEV)
    # oracle: host tokeniser ids -> fp32 text tower; Pillow-exact transform -> fp32 image tower; numpy float64 combine
    model = s2i.get_available_models()[s2i._create_model_cache_key("tiny-clip", DEV, props)]["model"]
    ids = torch.from_numpy(model.tokenizer([d["text"] for d in docs]))
    t_ref = O.clip_text_forward(sd, tcfg, ids)
    px = torch.from_numpy(np.stack([OP.clip_transform(np.asarray(d["image"]), 64) for d in docs]))
    i_ref = O.vit_forward(sd, vcfg, px)
    t_ref = (t_ref / t_ref.norm(dim=-1, keepdim=True)).numpy()
    i_ref = (i_ref / i_ref.norm(dim=-1, keepdim=True)).numpy()
    for j, d in enumerate(docs):
        assert _cos_err(emb[(d["_id"], "text")], t_ref[j]) < 1e-3 and _cos_err(emb[(d["_id"], "image")], i_ref[j]) < 1e-3
        ref = OC.combine_multimodal([t_ref[j], i_ref[j]], [weights["text"], weights["image"]], True)
        assert _cos_err(got[j], ref) < 1e-3
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)


def test_config5_fp8_towers_with_device_chunking(s2):
    s2i, root = s2
    props, sd, vcfg, tcfg = _tiny_clip(root)
    props8 = dict(props, enginePrecision="fp8")
    rng = np.random.default_rng(5)
    imgs = [Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for h, w in [(90, 120), (64, 64), (200, 70), (33, 47)]]
    # the model cache keys on the properties, so the fp8 model is a separate instance from the bf16 one
    s2i.vectorise("tiny-clip-fp8", imgs[:2], model_properties=props8, device=DEV, modality=s2i.Modality.IMAGE)
    model = s2i.get_available_models()[s2i._create_model_cache_key("tiny-clip-fp8", DEV, props8)]["model"]
    assert model.vision.precision == "fp8" and model.vision._fp8.calibrated
    emb, boxes = model.encode_image_chunks(imgs, 3, 3, False)
    assert emb.shape == (4, 10, 64) and boxes.shape == (4, 10, 4)
    for i, im in enumerate(imgs):
        patches, bbs = OP.chunk_image_simple(np.asarray(im), 3, 3, False)
        ref = O.vit_forward(sd, vcfg, torch.from_numpy(np.stack([OP.clip_transform(p, 64) for p in patches]))).numpy()
        assert np.allclose(boxes[i], np.asarray(bbs), rtol=1e-6)
        assert _cos_err(emb[i], ref) < 1e-3  # the north-star tolerance is global: the load-time bf16 / e4m3 block split (7e-4 budget vs bf16) holds it
    again, _ = model.encode_image_chunks(imgs, 3, 3, False)
    assert np.array_equal(emb, again)  # frozen scales: deterministic
    # text side of the same fp8 model, through the device tokeniser
    texts = ["a photo of a cat", "the quick brown fox jumps over the lazy dog"]
    out = np.asarray(s2i.vectorise("tiny-clip-fp8", texts, model_properties=props8, device=DEV))
    ref = O.clip_text_forward(sd, tcfg, torch.from_numpy(model.tokenizer(texts))).numpy()
    assert _cos_err(out, ref) < 1e-3
    from marqo_amd.s2_inference.errors import ModelLoadError
    with pytest.raises(ModelLoadError):
        s2i.vectorise("tiny-clip-bad", texts, model_properties=dict(props, enginePrecision="int4"), device=DEV)
