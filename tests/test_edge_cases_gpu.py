"""Edge cases of the hot path on the GPU (the cases the reference's tests exercise for this path: ragged / minimal /
maximal inputs, batch-of-one, chunked calls, concurrent callers), each against the fp32 CPU oracle."""
import threading

import numpy as np
import pytest
import torch

from oracle import towers as O

pytestmark = pytest.mark.gpu
COS_TOL = 1e-3


def _cos_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((1 - (a * b).sum(-1) / (a.norm(dim=-1) * b.norm(dim=-1))).max())


@pytest.fixture(scope="module")
def small_bert():
    from marqo_amd.engine import archs, towers
    cfg = O.BertConfig(vocab=2000, max_pos=512, width=128, layers=2, heads=2, mlp_dim=256)
    sd = O.synthetic_bert_state_dict(cfg, seed=11)
    arch = archs.BertArch(vocab=2000, max_pos=512, width=128, layers=2, heads=2, mlp_dim=256)
    return cfg, sd, towers.BertTower(arch, sd, "cuda:0"), towers.BertTower(arch, sd, "cuda:0", pooling="cls")


def test_bert_ragged_lengths_up_to_max_positions(small_bert):
    cfg, sd, mean_tower, cls_tower = small_bert
    g = torch.Generator().manual_seed(0)
    lens = [1, 2, 3, 17, 64, 65, 200, 511, 512]
    S = max(lens)
    ids = torch.zeros(len(lens), S, dtype=torch.int64)
    mask = torch.zeros(len(lens), S, dtype=torch.int64)
    for i, l in enumerate(lens):
        ids[i, :l] = torch.randint(1, 2000, (l,), generator=g)
        mask[i, :l] = 1
    ref = O.hf_encode(sd, cfg, ids, mask)
    assert _cos_err(mean_tower.encode_ids(ids, mask), ref) < COS_TOL
    cfg.pooling = "cls"
    try:
        assert _cos_err(cls_tower.encode_ids(ids, mask), O.hf_encode(sd, cfg, ids, mask)) < COS_TOL
    finally:
        cfg.pooling = "mean"
    with pytest.raises(ValueError):
        mean_tower.encode_ids(ids[:, :4], torch.tensor([[0, 1, 1, 1]] * len(lens)))  # not right-padded
    with pytest.raises(ValueError):
        mean_tower.encode_ids(ids[:1, :4], torch.zeros(1, 4, dtype=torch.int64))       # an empty sequence
    with pytest.raises(ValueError):
        mean_tower.encode_ids(torch.zeros(1, 600, dtype=torch.int64), torch.ones(1, 600, dtype=torch.int64))  # > max_pos


def test_clip_text_minimal_and_full_length():
    from marqo_amd.engine import archs, towers
    cfg = O.ClipTextConfig(vocab=1000, ctx=77, width=128, layers=2, heads=2, mlp_dim=256, out_dim=64)
    sd = O.synthetic_clip_text_state_dict(cfg, seed=5)
    tower = towers.ClipTextTower(archs.ClipTextArch(1000, 77, 128, 2, 2, 256, 64), sd, "cuda:0")
    ids = torch.zeros(3, 77, dtype=torch.int64)
    ids[0, :2] = torch.tensor([998, 999])                                  # SOT EOT only
    ids[1, :77] = torch.cat([torch.tensor([998]), torch.randint(1, 998, (75,)), torch.tensor([999])])  # full context
    ids[2, :5] = torch.tensor([998, 5, 6, 7, 999])
    ref = O.clip_text_forward(sd, cfg, ids)
    assert _cos_err(tower.encode_ids(ids), ref) < COS_TOL
    assert _cos_err(tower.encode_ids(ids, pack=False), ref) < COS_TOL
    assert tower.encode_ids(ids[:0]).shape == (0, 64)
    with pytest.raises(ValueError):
        tower.encode_ids(torch.zeros(1, 78, dtype=torch.int64))


def test_vit_batch_of_one_chunked_calls_and_threads(monkeypatch, tiled_gemm_only):
    from marqo_amd.engine import archs, towers
    cfg = O.VitConfig(image_size=64, patch_size=16, width=128, layers=2, heads=2, mlp_dim=256, out_dim=64)
    sd = O.synthetic_vit_state_dict(cfg, seed=7)
    arch = archs.VitArch(64, 16, 128, 2, 2, 256, 64)
    u8 = O.synthetic_images_u8(23, 64, seed=1)
    ref = O.vit_forward(sd, cfg, O.preprocess_u8_exact_size(u8))
    tower = towers.VitTower(arch, sd, "cuda:0")
    full = tower.encode_u8(u8.cuda())
    assert _cos_err(full, ref) < COS_TOL
    assert _cos_err(tower.encode_u8(u8[:1].cuda()), ref[:1]) < COS_TOL
    assert tower.encode_u8(u8[:0].cuda()).shape == (0, 64)
    # chunked: force 5 images per C-ABI call -> 5 calls, identical result
    monkeypatch.setattr(tower, "max_images_per_call", 5)
    assert torch.equal(tower.encode_u8(u8.cuda()), full)
    # un-normalised output really is un-normalised and proportional
    raw = tower.encode_u8(u8.cuda(), normalize=False)
    assert not torch.allclose(raw.norm(dim=-1), torch.ones(23, device="cuda"), atol=1e-2)
    assert _cos_err(raw, full) < 1e-6
    # wrong shapes / dtypes are rejected before anything is launched
    with pytest.raises(ValueError):
        tower.encode_u8(torch.zeros(2, 64, 64, 3))
    with pytest.raises(ValueError):
        tower.encode_u8(torch.zeros(2, 32, 32, 3, dtype=torch.uint8))
    # concurrent callers (FastAPI worker threads share one loaded model): per-tower lock + per-call outputs
    outs, errs = [None] * 8, []

    def work(i):
        try:
            outs[i] = tower.encode_u8(u8[i:i + 9].cuda()).cpu()
        except Exception as e:  # pragma: no cover
            errs.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs
    for i in range(8):
        assert torch.equal(outs[i], full[i:i + 9].cpu())


def test_no_cpu_fallback_and_bad_devices():
    from marqo_amd import _lib as L
    from marqo_amd.engine import archs, towers
    cfg = O.VitConfig(image_size=64, patch_size=16, width=128, layers=1, heads=2, mlp_dim=256, out_dim=64)
    sd = O.synthetic_vit_state_dict(cfg, seed=7)
    with pytest.raises(L.MarqoHipUnavailableError):
        towers.VitTower(archs.VitArch(64, 16, 128, 1, 2, 256, 64), sd, "cpu")
    with pytest.raises(ValueError):
        towers.VitTower(archs.VitArch(64, 16, 96, 1, 2, 256, 64), sd, "cuda:0")   # head dim != 64
    with pytest.raises(KeyError):
        towers.VitTower(archs.VitArch(64, 16, 128, 3, 2, 256, 64), sd, "cuda:0")  # checkpoint lacks layer 2
