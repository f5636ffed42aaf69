"""LayerNorm folding (csrc/gemm_epilogue.h; an opt-in variant, MARQO_AMD_LN_FOLD=1 + mq_tune("ln_fold", 1) — it measured
SLOWER than the separate LayerNorm kernels): the LayerNorm between a residual GEMM and the next GEMM is folded
into the two epilogues.  Checks the two epilogue modes against plain PyTorch fp32, the folded towers against the un-folded ones and the CPU
oracle, and the numerical behaviour on rows whose mean is several standard deviations away from zero."""
import pytest
import torch

from marqo_amd import _lib as L
from oracle import towers as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _tiled_family(tiled_gemm_only):
    """the folded epilogues live in the tiled kernels: compare against that family on the small shapes too"""
    yield


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _tune(key, value):
    L.check(L.load().mq_tune(key.encode(), value))


@pytest.mark.parametrize("M,N,K", [(300, 768, 768), (12800, 768, 3072), (1000, 512, 2048), (77, 1024, 64), (50, 192, 128)])
def test_producer_epilogue_writes_x_bf16_copy_and_row_partials(M, N, K):
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(M + N)
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g) * 3 + 1.5
    flags = L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32
    base = res.clone()
    L.check(lib.mq_gemm_bf16(A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), base.data_ptr(), base.data_ptr(), N, M, N, K, flags, _stream()))
    out = res.clone()
    nslots = (N + 63) // 64
    stats = torch.full((M, nslots, 2), float("nan"), device="cuda")
    xb = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    L.check(lib.mq_gemm_bf16_ln(A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), out.data_ptr(), out.data_ptr(), N, M, N, K,
                                flags | L.MQ_EPI_LN_STATS, stats.data_ptr(), xb.data_ptr(), 0, 1e-5, _stream()))
    assert torch.equal(out, base)                       # the fp32 residual stream is untouched by the extra outputs
    assert torch.equal(xb, out.to(torch.bfloat16))      # RNE bf16 copy
    pad = nslots * 64 - N
    o = torch.nn.functional.pad(out, (0, pad)).view(M, nslots, 64).double()
    assert torch.allclose(stats[..., 0].double(), o.sum(-1), rtol=1e-5, atol=1e-4)
    assert torch.allclose(stats[..., 1].double(), (o * o).sum(-1), rtol=1e-5, atol=1e-4)
    again = torch.empty_like(stats)
    out2 = res.clone()
    L.check(lib.mq_gemm_bf16_ln(A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), out2.data_ptr(), out2.data_ptr(), N, M, N, K,
                                flags | L.MQ_EPI_LN_STATS, again.data_ptr(), xb.data_ptr(), 0, 1e-5, _stream()))
    assert torch.equal(again, stats)                    # plain stores in a fixed order: deterministic


@pytest.mark.parametrize("act", ["none", "gelu", "quick"])
@pytest.mark.parametrize("M,N,K,mean", [(500, 2304, 768, 0.0), (12800, 3072, 768, 0.3), (333, 1536, 512, 3.0), (64, 256, 1024, -1.0)])
def test_consumer_epilogue_equals_layernorm_then_gemm(M, N, K, mean, act):
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(N + K)
    x = torch.randn(M, K, device="cuda", generator=g) * 2.0 + mean * 2.0          # row mean = `mean` standard deviations
    x[:, 7] += 40.0                                                                 # one massive-activation channel
    gam = 1 + 0.2 * torch.randn(K, device="cuda", generator=g)
    bet = 0.1 * torch.randn(K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = 0.1 * torch.randn(N, device="cuda", generator=g)
    eps = 1e-5
    ref = torch.nn.functional.layer_norm(x, (K,), gam, bet, eps) @ W.t() + b
    want = {"none": ref, "gelu": torch.nn.functional.gelu(ref), "quick": ref * torch.sigmoid(1.702 * ref)}[act]
    # what the loader precomputes (engine/towers.py _clip_blocks)
    wf = (W * gam.unsqueeze(0)).to(torch.bfloat16)
    colsum = wf.float().sum(1).contiguous()
    bf = (b + W @ bet).contiguous()
    # what the producer epilogue leaves behind
    xb = x.to(torch.bfloat16)
    nslots = (K + 63) // 64
    xs = torch.nn.functional.pad(x, (0, nslots * 64 - K)).view(M, nslots, 64)
    stats = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], dim=-1).contiguous()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    flags = L.MQ_EPI_BIAS | L.MQ_EPI_LN_APPLY | {"none": 0, "gelu": L.MQ_EPI_GELU, "quick": L.MQ_EPI_QUICKGELU}[act]
    L.check(lib.mq_gemm_bf16_ln(xb.data_ptr(), K, wf.data_ptr(), K, bf.data_ptr(), 0, out.data_ptr(), N, M, N, K, flags, stats.data_ptr(), 0,
                                colsum.data_ptr(), eps, _stream()))
    # the un-folded engine path on the same inputs, for scale: LN -> bf16 -> GEMM
    h = torch.nn.functional.layer_norm(x, (K,), gam, bet, eps).to(torch.bfloat16)
    unf = torch.empty_like(out)
    flags_u = L.MQ_EPI_BIAS | {"none": 0, "gelu": L.MQ_EPI_GELU, "quick": L.MQ_EPI_QUICKGELU}[act]
    Wb = W.to(torch.bfloat16)
    L.check(lib.mq_gemm_bf16(h.data_ptr(), K, Wb.data_ptr(), K, b.data_ptr(), 0, unf.data_ptr(), N, M, N, K, flags_u, _stream()))
    scale = want.abs().max().item()
    err_f = (out.float() - want).abs().max().item() / scale
    err_u = (unf.float() - want).abs().max().item() / scale
    rms_f = ((out.float() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
    rms_u = ((unf.float() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
    print(f"M={M} N={N} K={K} mean={mean} {act}: folded max {err_f:.2e} rms {rms_f:.2e} | unfolded max {err_u:.2e} rms {rms_u:.2e}")
    assert err_f < 2.5e-2 and rms_f < 8e-3
    assert rms_f < 3.0 * rms_u + 1e-3   # same error class as LN -> bf16 -> GEMM, also at |mean| = 3 sigma


def test_folded_towers_match_unfolded_and_oracle(monkeypatch):
    from marqo_amd.engine import archs, towers
    monkeypatch.setattr(towers, "LN_FOLD", True)  # build the folded tensors (off by default)
    varch = archs.VitArch(224, 32, 768, 4, 12, 3072, 512)
    cfg = O.VitConfig(224, 32, 768, 4, 12, 3072, 512)
    sd = O.synthetic_vit_state_dict(cfg, seed=7)
    u8 = O.synthetic_images_u8(12, 224, seed=8)
    ref = O.vit_forward(sd, cfg, O.preprocess_u8_exact_size(u8))
    tower = towers.VitTower(varch, sd, "cuda")
    assert tower._blocks[0].qkv_wf and tower._blocks[0].fc1_sf  # folded tensors are present
    cos = lambda a, b: float((1 - torch.nn.functional.cosine_similarity(a.double().cpu(), b.double().cpu(), dim=-1)).max())
    try:
        _tune("ln_fold", 1)
        folded = tower.encode_u8(u8.cuda())
        assert torch.equal(folded, tower.encode_u8(u8.cuda()))  # deterministic
        _tune("ln_fold", 0)
        plain = tower.encode_u8(u8.cuda())
    finally:
        _tune("ln_fold", 0)
    e_f, e_p, e_fp = cos(folded, ref), cos(plain, ref), cos(folded, plain)
    print(f"ViT-B/32 x4 layers: 1-cos vs oracle folded {e_f:.2e} plain {e_p:.2e}; folded vs plain {e_fp:.2e}")
    assert e_f < 3e-4 and e_p < 3e-4 and e_fp < 1e-4
    # text tower (causal, packed ragged sequences)
    tarch = archs.ClipTextArch(49408, 77, 512, 3, 8, 2048, 512)
    tcfg = O.ClipTextConfig(49408, 77, 512, 3, 8, 2048, 512)
    sdt = O.synthetic_clip_text_state_dict(tcfg, seed=9)
    ids = O.synthetic_clip_ids(10, seed=10)
    tt = towers.ClipTextTower(tarch, sdt, "cuda")
    reft = O.clip_text_forward(sdt, tcfg, ids)
    try:
        _tune("ln_fold", 1)
        f = tt.encode_ids(ids)
        _tune("ln_fold", 0)
        p = tt.encode_ids(ids)
    finally:
        _tune("ln_fold", 0)
    assert cos(f, reft) < 3e-4 and cos(p, reft) < 3e-4 and cos(f, p) < 1e-4
