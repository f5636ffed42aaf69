"""The golden embedding vectors the REFERENCE'S OWN TESTS hold for this path (tests/golden/ref_vectors.npz, lifted digit for digit
from /root/reference/tests/core/inference/... by tests/golden/make_ref_vectors.py), wired as weights-gated tests.

They pin REAL checkpoints (intfloat/e5-base-v2, sentence-transformers/nli-bert-base-cls-pooling, Marqo/marqo-fashionCLIP,
Marqo/marqo-fashionSigLIP) which cannot be downloaded here (no network): each test runs when the checkpoint directory is mounted under
`$MARQO_AMD_MODEL_DIR` (HF repos as `hf/<org>/<repo>/{config.json, model.safetensors, vocab.txt}`; open_clip hub repos as
`hf-hub/<org>/<repo>/{open_clip_model.safetensors, open_clip_config.json, tokenizer files}`; the test image as `fashion-hippo.png`) and
skips otherwise.  With weights mounted:
  * the fp32 CPU oracle must meet the REFERENCE'S OWN tolerance (test_hugging_face_model.py:631-634: ||emb - gold||_2 / len(emb) < 1e-4 with
    len(emb) = 1 row; test_marqo_fashion_clip.py:581-589: ||emb - gold||_2 / D < 1e-4) — that turns "parity unpinned" for real weights green;
  * the HIP bf16 path must meet the north-star tolerance, 1 - cos < 1e-3 against the same vectors (and the reference's fashion-clip bound).
Without weights only the fixture's self-consistency runs (unit norms, dims, the reference's "different models differ by > 1" check)."""
import json
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load():
    z = np.load(os.path.join(GOLDEN, "ref_vectors.npz"))
    meta = json.loads(bytes(z["__meta__"]).decode())
    return {k: z[k] for k in z.files if k != "__meta__"}, meta


VEC, META = _load()


def _model_root():
    from marqo_amd.engine import checkpoint
    return checkpoint.model_dir()


def _hf_dir(repo):
    from marqo_amd.engine import checkpoint
    return checkpoint.find_hf_dir(repo)


def _clip_ckpt(repo):
    from marqo_amd.engine import checkpoint
    return checkpoint.find_open_clip_checkpoint("hf-hub:" + repo, None)


def _hippo():
    p = os.path.join(_model_root(), "fashion-hippo.png")
    return p if os.path.isfile(p) else None


def test_fixture_is_the_references_data():
    assert set(VEC) == {"E5_BASE_V2_MODEL_EMBEDDINGS", "NLI_BERT_BASE_CLS_MODEL_EMBEDDINGS", "FASHIONCLIP_IMAGE_EMBEDDING",
                        "FASHIONCLIP_TEXT_EMBEDDING", "SiGLIP_IMAGE_EMBEDDING", "SiGLIP_TEXT_EMBEDDING"}
    for k, v in VEC.items():
        assert v.ndim == 1 and v.shape[0] == META[k]["dim"] and abs(np.linalg.norm(v) - 1.0) < 2e-4, k   # every one is L2-normalised
        assert META[k]["source"].startswith("tests/core/inference/")
    assert VEC["E5_BASE_V2_MODEL_EMBEDDINGS"].shape == (768,) and VEC["FASHIONCLIP_IMAGE_EMBEDDING"].shape == (512,)
    # test_hugging_face_model.py:760-768: two different models must differ by more than 1 in L2
    assert np.linalg.norm(VEC["NLI_BERT_BASE_CLS_MODEL_EMBEDDINGS"] - VEC["E5_BASE_V2_MODEL_EMBEDDINGS"]) > 1
    # registry dims of the models these vectors pin
    from marqo_amd.s2_inference.model_registry import load_model_properties
    models = load_model_properties()["models"]
    assert models["hf/e5-base-v2"]["dimensions"] == 768
    assert models["Marqo/marqo-fashionCLIP"]["dimensions"] == 512 and models["Marqo/marqo-fashionSigLIP"]["dimensions"] == 768


HF_CASES = [("E5_BASE_V2_MODEL_EMBEDDINGS", {"name": "intfloat/e5-base-v2", "dimensions": 768, "tokens": 512, "type": "hf"}),
            ("NLI_BERT_BASE_CLS_MODEL_EMBEDDINGS", {"name": "sentence-transformers/nli-bert-base-cls-pooling", "dimensions": 768, "type": "hf"})]


@pytest.mark.parametrize("key,props", HF_CASES)
def test_oracle_meets_the_references_own_tolerance_on_real_hf_weights(key, props):
    d = _hf_dir(props["name"])
    if d is None:
        pytest.skip(f"{props['name']} is not mounted under $MARQO_AMD_MODEL_DIR (no network here)")
    from marqo_amd.engine import archs, checkpoint
    from marqo_amd.engine.tokenizers import WordPieceTokenizer
    from oracle import towers as O
    cfg, sd = checkpoint.load_hf_dir(d)
    a = archs.bert_arch_from_hf_config(cfg)
    pooling = props.get("poolingMethod") or checkpoint.read_pooling_config(d) or "mean"
    assert pooling == META[key]["pooling"]
    if "embeddings.word_embeddings.weight" not in sd:
        sd = {k.split(".", 1)[1]: v for k, v in sd.items() if k.startswith(("bert.", "roberta."))}
    sd = {k: v.float() for k, v in sd.items()}
    tok = WordPieceTokenizer(d)([META[key]["content"]], max_length=props.get("tokens", 128))
    ocfg = O.BertConfig(vocab=a.vocab, max_pos=a.max_pos, width=a.width, layers=a.layers, heads=a.heads, mlp_dim=a.mlp_dim, ln_eps=a.ln_eps,
                        pooling=pooling, pos_offset=a.pos_offset)
    emb = O.hf_encode(sd, ocfg, torch.from_numpy(tok["input_ids"]), torch.from_numpy(tok["attention_mask"])).numpy()
    difference = np.linalg.norm(emb - VEC[key]) / len(emb)          # the reference's own expression (len(emb) == 1)
    assert difference < 1e-4, difference


@pytest.mark.gpu
@pytest.mark.parametrize("key,props", HF_CASES)
def test_hip_path_on_real_hf_weights(key, props):
    if _hf_dir(props["name"]) is None:
        pytest.skip(f"{props['name']} is not mounted under $MARQO_AMD_MODEL_DIR (no network here)")
    from marqo_amd.s2_inference import s2_inference as s2
    emb = np.asarray(s2.vectorise(props["name"], [META[key]["content"]], model_properties=props, device="cuda:0"))
    gold = VEC[key]
    cos = float((emb[0] * gold).sum() / (np.linalg.norm(emb[0]) * np.linalg.norm(gold)))
    assert 1 - cos < 1e-3, (1 - cos, np.linalg.norm(emb - gold))


CLIP_CASES = [("Marqo/marqo-fashionCLIP", "FASHIONCLIP"), ("Marqo/marqo-fashionSigLIP", "SiGLIP")]


@pytest.mark.gpu
@pytest.mark.parametrize("repo,prefix", CLIP_CASES)
def test_hip_path_on_real_fashion_clip_weights(repo, prefix):
    if _clip_ckpt(repo) is None:
        pytest.skip(f"{repo} is not mounted under $MARQO_AMD_MODEL_DIR (no network here)")
    from marqo_amd.s2_inference import s2_inference as s2
    props = s2.get_model_properties_from_registry(repo)
    text = np.squeeze(np.asarray(s2.vectorise(repo, "a hat", model_properties=props, device="cuda:0")))
    gold = VEC[f"{prefix}_TEXT_EMBEDDING"]
    assert np.linalg.norm(text - gold) / len(text) < 1e-4           # the reference's own expression and bound
    assert 1 - float(text @ gold / np.linalg.norm(text) / np.linalg.norm(gold)) < 1e-3
    if _hippo() is None:
        pytest.skip("fashion-hippo.png is not mounted under $MARQO_AMD_MODEL_DIR")
    from PIL import Image
    img = np.squeeze(np.asarray(s2.vectorise(repo, [Image.open(_hippo())], model_properties=props, device="cuda:0", modality=s2.Modality.IMAGE)))
    gold = VEC[f"{prefix}_IMAGE_EMBEDDING"]
    assert np.linalg.norm(img - gold) / len(img) < 1e-4
    assert 1 - float(img @ gold / np.linalg.norm(img) / np.linalg.norm(gold)) < 1e-3
