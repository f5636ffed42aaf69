"""`mq_attention_proj` (ABI 13, csrc/attn_proj.hip): attention + out-projection + residual + the LayerNorm statistics behind it as ONE launch for the
short fixed-length sequences of the ViT-B/32 image tower — against plain PyTorch fp32, against the three launches it replaces (bit-identical rows), and
inside the tower.  Reference arithmetic: open_clip's ResidualAttentionBlock (x = x + out_proj(attention(ln_1(x)))), reached from
/root/reference/src/marqo/core/inference/embedding_models/open_clip_model.py:249-266."""
import pytest
import torch

from marqo_amd import _lib as L

pytestmark = pytest.mark.gpu

W, HEADS = 768, 12


def _s():
    return torch.cuda.current_stream().cuda_stream


def _inputs(nseq, T, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rows = nseq * T
    qkv = (torch.randn(rows, 3 * W, device="cuda", generator=g) * 1.3).to(torch.bfloat16)
    qkv[:, 5] += 4.0                                              # an outlier channel in q: peaky softmax rows
    wo = (torch.randn(W, W, device="cuda", generator=g) / W ** 0.5).to(torch.bfloat16)
    bias = 0.1 * torch.randn(W, device="cuda", generator=g)
    x0 = (torch.randn(rows, W, device="cuda", generator=g) * 2 + 0.5).to(torch.bfloat16)
    return qkv, wo, bias, x0


def _three_launches(qkv, wo, bias, x0, nseq, T, eps):
    """mq_attention -> mq_gemm_bf16_rs (bias + bf16 residual in place + partial row sums) -> mq_row_stats_finalize"""
    lib = L.load()
    rows = nseq * T
    a = torch.empty(rows, W, device="cuda", dtype=torch.bfloat16)
    L.check(lib.mq_attention(qkv.data_ptr(), a.data_ptr(), None, nseq, T, T, W, HEADS, L.MQ_MASK_NONE, _s()))
    x = x0.clone()
    ns = (W + 63) // 64
    part = torch.empty(ns, rows, 2, device="cuda")
    L.check(lib.mq_gemm_bf16_rs(a.data_ptr(), W, wo.data_ptr(), W, bias.data_ptr(), x.data_ptr(), x.data_ptr(), W, rows, W, W, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL,
                                part.data_ptr(), _s()))
    stats = torch.empty(rows, 2, device="cuda")
    L.check(lib.mq_row_stats_finalize(part.data_ptr(), ns, stats.data_ptr(), rows, W, eps, _s()))
    return a, x, stats


def _fused(qkv, wo, bias, x0, nseq, T, eps, with_stats=True):
    lib = L.load()
    x = x0.clone()
    stats = torch.full((nseq * T, 2), float("nan"), device="cuda")
    L.check(lib.mq_attention_proj(qkv.data_ptr(), wo.data_ptr(), bias.data_ptr(), x.data_ptr(), stats.data_ptr() if with_stats else None, nseq, T, W, HEADS, eps, None, 0, None, 0, _s()))
    return x, stats


@pytest.mark.parametrize("nseq,T", [(256, 50), (3, 50), (300, 50), (5, 64), (7, 17), (2, 1), (64, 49), (9, 33)])
def test_one_launch_equals_the_three_it_replaces_and_fp32_torch(nseq, T):
    eps = 1e-5
    qkv, wo, bias, x0 = _inputs(nseq, T, 100 * nseq + T)
    a, x3, s3 = _three_launches(qkv, wo, bias, x0, nseq, T, eps)
    x1, s1 = _fused(qkv, wo, bias, x0, nseq, T, eps)
    torch.cuda.synchronize()
    # the rows: the same operations in the same order -> the same bits
    assert torch.equal(x1.view(torch.int16), x3.view(torch.int16))
    # the statistics: the partial-sum path's bits too (same per-lane chains, same lane folds, same slot order: an embedding may not depend on which path its batch took)
    assert torch.equal(s1, s3)
    xd = x1.double()
    mean, rstd = xd.mean(1), 1.0 / torch.sqrt(xd.var(1, unbiased=False) + eps)
    assert torch.allclose(s1[:, 0].double(), mean, rtol=0, atol=2e-6 * float(xd.abs().max()))
    assert torch.allclose(s1[:, 1].double(), rstd, rtol=2e-5, atol=0)
    # ... and plain PyTorch fp32 (bf16 rounding of P and of the attention output are the kernel's own)
    q, k, v = (qkv.float().view(nseq, T, 3, HEADS, 64).permute(2, 0, 3, 1, 4))
    o = torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1) @ v
    o = o.permute(0, 2, 1, 3).reshape(nseq * T, W)
    want = x0.float() + o @ wo.float().t() + bias
    err = (x1.float() - want).abs().max() / want.abs().max()
    assert float(err) < 1.5e-2, float(err)
    cos = torch.nn.functional.cosine_similarity(x1.float(), want, dim=-1)
    assert float((1 - cos).max()) < 3e-5


def test_without_statistics_and_repeatable():
    qkv, wo, bias, x0 = _inputs(40, 50, 7)
    xa, sa = _fused(qkv, wo, bias, x0, 40, 50, 1e-6)
    xb, sb = _fused(qkv, wo, bias, x0, 40, 50, 1e-6)
    xc, sc = _fused(qkv, wo, bias, x0, 40, 50, 1e-6, with_stats=False)
    torch.cuda.synchronize()
    assert torch.equal(xa.view(torch.int16), xb.view(torch.int16)) and torch.equal(sa, sb)
    assert torch.equal(xa.view(torch.int16), xc.view(torch.int16)) and bool(torch.isnan(sc).all())


def test_an_image_has_the_same_bits_wherever_it_stands_in_the_batch():
    qkv, wo, bias, x0 = _inputs(70, 50, 11)
    x_all, s_all = _fused(qkv, wo, bias, x0, 70, 50, 1e-5)
    perm = torch.randperm(70, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    rows = (perm[:, None] * 50 + torch.arange(50, device="cuda")[None, :]).reshape(-1)
    x_p, s_p = _fused(qkv[rows].contiguous(), wo, bias, x0[rows].contiguous(), 70, 50, 1e-5)
    torch.cuda.synchronize()
    assert torch.equal(x_p.view(torch.int16), x_all[rows].view(torch.int16)) and torch.equal(s_p, s_all[rows])


def test_shapes_it_does_not_take_are_refused():
    lib = L.load()
    assert lib.mq_attention_proj_ok(256, 50, 768, 12) == 1
    for nseq, T, w, h in [(256, 65, 768, 12), (256, 50, 1024, 16), (0, 50, 768, 12), (4, 0, 768, 12), (4, 50, 768, 8)]:
        assert lib.mq_attention_proj_ok(nseq, T, w, h) == 0
    qkv, wo, bias, x0 = _inputs(2, 50, 1)
    assert lib.mq_attention_proj(qkv.data_ptr(), wo.data_ptr(), bias.data_ptr(), x0.data_ptr(), None, 2, 65, W, HEADS, 1e-5, None, 0, None, 0, _s()) == -1      # MQ_ERR_INVALID


def test_vit_b32_tower_with_and_without_the_fused_launch():
    """the tower (bf16 residual stream, folded LayerNorms) with the block's attention half as one launch vs as three: bit-identical embeddings, on the oracle"""
    from marqo_amd.engine import archs, towers
    from oracle import towers as O
    lib = L.load()
    varch, _ = archs.resolve_open_clip("ViT-B-32")
    cfg = O.VitConfig(varch.image_size, varch.patch_size, varch.width, varch.layers, varch.heads, varch.mlp_dim, varch.out_dim)
    sd = O.synthetic_vit_state_dict(cfg, seed=0)
    u8 = O.synthetic_images_u8(72, varch.image_size, seed=2).to("cuda:0")
    tower = towers.VitTower(varch, sd, "cuda:0")
    try:
        L.check(lib.mq_tune(b"attn_proj", 0))
        three = tower.encode_u8(u8).cpu()
        L.check(lib.mq_tune(b"attn_proj", 1))
        one = tower.encode_u8(u8).cpu()
        one_small = tower.encode_u8(u8[:5]).cpu()
    finally:
        L.check(lib.mq_tune(b"attn_proj", 128))
    assert torch.equal(one, three)          # rows AND statistics carry the three launches' bits: so do the embeddings
    ref = O.vit_forward(sd, cfg, O.preprocess_u8_exact_size(u8[:6].cpu()))
    for got in (one[:6], three[:6]):
        c = torch.nn.functional.cosine_similarity(got.double(), ref.double(), dim=-1)
        assert float((1 - c).max()) < 1e-3
    if tower.residual_stream == "bf16":     # (the fused launch rides on the bf16 stream; a tower tuned to fp32 never takes it)
        c5 = torch.nn.functional.cosine_similarity(one_small.double(), one[:5].double(), dim=-1)
        assert float((1 - c5).max()) < 2e-4
