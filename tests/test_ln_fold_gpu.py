"""LayerNorm folding on the bf16 residual stream (csrc/gemm_epilogue.h, MQ_EPI_LN_APPLY; round 4, default for the pre-LN towers): a one-pass statistics
kernel (mq_row_stats: (mean, rstd) per row, the LayerNorm's own two-pass arithmetic) and the QKV / fc1 GEMM that reads the UN-normalised stream and
applies the LayerNorm in its epilogue — no LayerNorm launch, no normalised copy.  Checks both kernels against plain PyTorch fp32 (LayerNorm -> Linear
[-> activation]) incl. rows whose mean is several standard deviations from zero and a massive-activation channel, determinism, and the folded towers
against the un-folded ones and the CPU oracle.
Reference arithmetic: open_clip ResidualAttentionBlock ln_1 -> attn.in_proj, ln_2 -> mlp.c_fc (reached from
/root/reference/src/marqo/core/inference/embedding_models/open_clip_model.py:249-266)."""
import pytest
import torch

from marqo_amd import _lib as L
from oracle import towers as O

pytestmark = pytest.mark.gpu


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _tune(key, value):
    L.check(L.load().mq_tune(key.encode(), value))


def _ln_gemm(lib, xb, wf, bf, colsum, out, flags, eps):
    M, K = xb.shape
    N = wf.shape[0]
    stats = torch.empty(M, 2, device="cuda")
    L.check(lib.mq_row_stats(xb.data_ptr(), stats.data_ptr(), M, K, eps, _stream()))
    L.check(lib.mq_gemm_bf16_ln(xb.data_ptr(), K, wf.data_ptr(), K, bf.data_ptr(), colsum.data_ptr(), stats.data_ptr(), out.data_ptr(), N, M, N, K, flags, _stream()))
    return stats


def _folded(W, b, gam, bet):
    """what the loader precomputes (engine/towers.py::_clip_blocks)"""
    wf = (W * gam.unsqueeze(0)).to(torch.bfloat16)
    return wf, (b + W @ bet).contiguous(), wf.float().sum(1).contiguous()


@pytest.mark.parametrize("act", ["none", "gelu", "quick"])
@pytest.mark.parametrize("M,N,K,mean", [(500, 2304, 768, 0.0), (12800, 3072, 768, 0.3), (333, 1536, 512, 3.0), (64, 256, 1024, -1.0), (161, 132, 64, 0.5),
                                        (4100, 4096, 1024, 0.1), (700, 1664, 1664, 0.2), (9000, 1536, 512, 0.0)])
def test_ln_apply_gemm_equals_layernorm_then_gemm(M, N, K, mean, act):
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(N + K)
    x = torch.randn(M, K, device="cuda", generator=g) * 2.0 + mean * 2.0          # row mean = `mean` standard deviations
    x[:, 7] += 40.0                                                                 # one massive-activation channel
    xb = x.to(torch.bfloat16)                                                       # the bf16 stream IS the operand: statistics are those of the rounded rows
    x = xb.float()
    gam = 1 + 0.2 * torch.randn(K, device="cuda", generator=g)
    bet = 0.1 * torch.randn(K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = 0.1 * torch.randn(N, device="cuda", generator=g)
    eps = 1e-5
    ref = torch.nn.functional.layer_norm(x, (K,), gam, bet, eps) @ W.t() + b
    want = {"none": ref, "gelu": torch.nn.functional.gelu(ref), "quick": ref * torch.sigmoid(1.702 * ref)}[act]
    wf, bf, colsum = _folded(W, b, gam, bet)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    flags = L.MQ_EPI_BIAS | {"none": 0, "gelu": L.MQ_EPI_GELU, "quick": L.MQ_EPI_QUICKGELU}[act]
    stats = _ln_gemm(lib, xb, wf, bf, colsum, out, flags, eps)
    # the statistics kernel against fp64 on the rounded rows
    xd = x.double()
    assert torch.allclose(stats[:, 0].double(), xd.mean(1), rtol=1e-5, atol=1e-5)
    assert torch.allclose(stats[:, 1].double(), 1.0 / torch.sqrt(xd.var(1, unbiased=False) + eps), rtol=2e-5)
    again = torch.empty_like(out)
    _ln_gemm(lib, xb, wf, bf, colsum, again, flags, eps)
    assert torch.equal(out.view(torch.int16), again.view(torch.int16))              # fixed reduction order: deterministic
    # the un-folded engine path on the same inputs, for scale: LN -> bf16 -> GEMM
    h = torch.nn.functional.layer_norm(x, (K,), gam, bet, eps).to(torch.bfloat16)
    unf = torch.empty_like(out)
    Wb = W.to(torch.bfloat16)
    L.check(lib.mq_gemm_bf16(h.data_ptr(), K, Wb.data_ptr(), K, b.data_ptr(), 0, unf.data_ptr(), N, M, N, K, flags, _stream()))
    scale = want.abs().max().item()
    err_f = (out.float() - want).abs().max().item() / scale
    rms_f = ((out.float() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
    rms_u = ((unf.float() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
    print(f"M={M} N={N} K={K} mean={mean} {act}: folded max {err_f:.2e} rms {rms_f:.2e} | unfolded rms {rms_u:.2e}")
    assert err_f < 2.5e-2 and rms_f < 8e-3
    assert rms_f < 3.0 * rms_u + 1e-3   # same error class as LN -> bf16 -> GEMM, also at |mean| = 3 sigma


def test_row_statistics_do_not_depend_on_the_tile_a_row_falls_into():
    """a row's result must not depend on which other rows share its call (requests are merged and split freely): same bits for a row alone-ish and
    inside a big batch — rows at different tile positions, ragged last tile"""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(5)
    K, N = 768, 2304
    xb = (torch.randn(1500, K, device="cuda", generator=g) * 1.5 + 0.4).to(torch.bfloat16)
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    wf, bf, colsum = _folded(W, 0.1 * torch.randn(N, device="cuda", generator=g), 1 + 0.1 * torch.randn(K, device="cuda", generator=g),
                             0.1 * torch.randn(K, device="cuda", generator=g))
    big = torch.empty(1500, N, device="cuda", dtype=torch.bfloat16)
    _ln_gemm(lib, xb, wf, bf, colsum, big, L.MQ_EPI_BIAS, 1e-5)
    part = torch.empty(700, N, device="cuda", dtype=torch.bfloat16)
    sub = xb[333:1033].contiguous()
    _ln_gemm(lib, sub, wf, bf, colsum, part, L.MQ_EPI_BIAS, 1e-5)
    assert torch.equal(part.view(torch.int16), big[333:1033].view(torch.int16))


@pytest.mark.parametrize("M,N,K", [(12800, 768, 768), (12800, 768, 3072), (1500, 512, 2048), (333, 1024, 1024), (700, 1664, 1664), (130, 72, 64), (9000, 1280, 5120)])
def test_residual_gemm_leaves_the_row_sums_the_next_layernorm_needs(M, N, K):
    """MQ_EPI_ROW_STATS: the out-proj / fc2 GEMM (bf16 read-modify-write of the stream) writes (sum, sum of squares) of every row's ROUNDED new values per
    64-column slot (slot-major: [slots][M]); mq_row_stats_finalize turns them into the (mean, rstd) mq_row_stats would have read back — the statistics pass over the stream goes"""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(M + N)
    a = (torch.randn(M, K, device="cuda", generator=g)).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = 0.1 * torch.randn(N, device="cuda", generator=g)
    x0 = (torch.randn(M, N, device="cuda", generator=g) * 2 + 0.5)
    x0[:, 5] += 60.0
    x0 = x0.to(torch.bfloat16)
    flags = L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL
    plain = x0.clone()
    L.check(lib.mq_gemm_bf16(a.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), plain.data_ptr(), plain.data_ptr(), N, M, N, K, flags, _stream()))
    nslots = (N + 63) // 64
    x = x0.clone()
    part = torch.full((nslots, M, 2), float("nan"), device="cuda")      # slot-major (ABI 12)
    L.check(lib.mq_gemm_bf16_rs(a.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), x.data_ptr(), x.data_ptr(), N, M, N, K, flags, part.data_ptr(), _stream()))
    assert torch.equal(x.view(torch.int16), plain.view(torch.int16))                # the stream itself: same bits with and without the by-product
    assert not torch.isnan(part).any()                                              # every (row, slot) written
    xd = x.double()
    pad = torch.zeros(M, nslots * 64, device="cuda", dtype=torch.float64)
    pad[:, :N] = xd
    pad = pad.view(M, nslots, 64)
    assert torch.allclose(part[..., 0].double().t(), pad.sum(-1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(part[..., 1].double().t(), pad.pow(2).sum(-1), rtol=1e-5, atol=1e-3)
    again = torch.empty_like(part)
    x2 = x0.clone()
    L.check(lib.mq_gemm_bf16_rs(a.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), x2.data_ptr(), x2.data_ptr(), N, M, N, K, flags, again.data_ptr(), _stream()))
    assert torch.equal(part, again)                                                 # fixed order: deterministic
    if N % 8 == 0 and N <= 2048:
        eps = 1e-5
        fin = torch.empty(M, 2, device="cuda")
        L.check(lib.mq_row_stats_finalize(part.data_ptr(), nslots, fin.data_ptr(), M, N, eps, _stream()))
        ref = torch.empty(M, 2, device="cuda")
        L.check(lib.mq_row_stats(x.data_ptr(), ref.data_ptr(), M, N, eps, _stream()))
        assert torch.allclose(fin[:, 0], ref[:, 0], rtol=1e-5, atol=1e-5)
        assert torch.allclose(fin[:, 1], ref[:, 1], rtol=1e-4)                      # one-pass variance in fp32 vs the statistics kernel's two passes
        assert torch.allclose(fin[:, 1].double(), 1.0 / torch.sqrt(xd.var(1, unbiased=False) + eps), rtol=1e-4)


@pytest.mark.parametrize("plan", ["default", "row_split", "mt2"])
@pytest.mark.parametrize("M,N,K", [(12800, 768, 768), (12800, 768, 3072), (16448, 1024, 4096), (1500, 512, 2048), (333, 1024, 1024), (700, 1664, 1664), (130, 72, 64), (4099, 1280, 128)])
def test_in_launch_finalise_gives_the_finalise_kernels_bits(M, N, K, plan, tiled_gemm_only):
    """round 6, mq_gemm_bf16_rsf: the residual GEMM finalises the row statistics inside its own launch (the last wave to arrive at a row band's counter sums
    the band's partials) — the same (mean, rstd) bits as mq_gemm_bf16_rs + mq_row_stats_finalize, the same stream and partials, counters left at zero, for
    every tile plan (ragged M and N, the big-tile row split, the smallest tile), launch after launch on the same counters, weight prefetch carried"""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(M * 3 + N)
    a = (torch.randn(M, K, device="cuda", generator=g)).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = 0.1 * torch.randn(N, device="cuda", generator=g)
    x0 = (torch.randn(M, N, device="cuda", generator=g) * 2 + 0.5).to(torch.bfloat16)
    flags = L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL
    nslots = (N + 63) // 64
    eps = 1e-5
    pf = torch.randn(3 << 20, device="cuda")       # 12 MB "weights of the next GEMM"
    try:
        _tune("rs_finalize", 1)
        if plan == "row_split":
            _tune("gemm_nh", 4)
        elif plan == "mt2":
            _tune("gemm_mt", 2)
        x = x0.clone()
        part = torch.full((nslots, M, 2), float("nan"), device="cuda")      # slot-major (ABI 12)
        L.check(lib.mq_gemm_bf16_rs(a.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), x.data_ptr(), x.data_ptr(), N, M, N, K, flags, part.data_ptr(), _stream()))
        want = torch.empty(M, 2, device="cuda")
        L.check(lib.mq_row_stats_finalize(part.data_ptr(), nslots, want.data_ptr(), M, N, eps, _stream()))
        ctr = torch.zeros(int(lib.mq_gemm_band_counters(M)), device="cuda", dtype=torch.int32)
        for rep in range(3):
            x2 = x0.clone()
            part2 = torch.full((nslots, M, 2), float("nan"), device="cuda")
            got = torch.full((M, 2), float("nan"), device="cuda")
            L.check(lib.mq_gemm_bf16_rsf(a.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), x2.data_ptr(), x2.data_ptr(), N, M, N, K, flags, part2.data_ptr(), got.data_ptr(), eps,
                                         ctr.data_ptr(), pf.data_ptr(), pf.numel() * 4, pf.data_ptr(), 4096, _stream()))
            assert torch.equal(x2.view(torch.int16), x.view(torch.int16)) and torch.equal(part2, part)
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (rep, int((got != want).sum()))
            assert int(ctr.abs().sum()) == 0
        # the two-launch form behind the same entry point (mq_tune rs_finalize = 0; no counters)
        for kw in (dict(rs=0, ctr=ctr.data_ptr()), dict(rs=1, ctr=0)):
            _tune("rs_finalize", kw["rs"])
            x3 = x0.clone()
            got = torch.full((M, 2), float("nan"), device="cuda")
            L.check(lib.mq_gemm_bf16_rsf(a.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), x3.data_ptr(), x3.data_ptr(), N, M, N, K, flags, part.data_ptr(), got.data_ptr(), eps,
                                         kw["ctr"], 0, 0, 0, 0, _stream()))
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)) and torch.equal(x3.view(torch.int16), x.view(torch.int16))
    finally:
        _tune("gemm_nh", 0); _tune("gemm_mt", 0); _tune("rs_finalize", 0)


def test_towers_give_the_same_bits_with_the_finalise_in_or_behind_the_launch():
    from marqo_amd.engine import archs, towers
    varch = archs.VitArch(224, 32, 768, 4, 12, 3072, 512)
    cfg = O.VitConfig(224, 32, 768, 4, 12, 3072, 512)
    sd = O.synthetic_vit_state_dict(cfg, seed=17)
    u8 = O.synthetic_images_u8(64, 224, seed=18).cuda()
    tower = towers.VitTower(varch, sd, "cuda")
    tarch = archs.ClipTextArch(49408, 77, 512, 3, 8, 2048, 512)
    sdt = O.synthetic_clip_text_state_dict(O.ClipTextConfig(49408, 77, 512, 3, 8, 2048, 512), seed=19)
    ids = O.synthetic_clip_ids(200, seed=20)
    tt = towers.ClipTextTower(tarch, sdt, "cuda")
    try:
        _tune("rs_finalize", 1)
        a1, t1 = tower.encode_u8(u8), tt.encode_ids(ids)
        assert torch.equal(a1, tower.encode_u8(u8)) and torch.equal(t1, tt.encode_ids(ids))
        _tune("rs_finalize", 0)
        a0, t0 = tower.encode_u8(u8), tt.encode_ids(ids)
    finally:
        _tune("rs_finalize", 0)
    assert torch.equal(a1, a0) and torch.equal(t1, t0)


def test_folded_towers_match_unfolded_and_oracle():
    from marqo_amd.engine import archs, towers
    assert towers.LN_FOLD
    varch = archs.VitArch(224, 32, 768, 4, 12, 3072, 512)
    cfg = O.VitConfig(224, 32, 768, 4, 12, 3072, 512)
    sd = O.synthetic_vit_state_dict(cfg, seed=7)
    u8 = O.synthetic_images_u8(24, 224, seed=8)     # 1 200 rows: the tiled (folded) GEMMs run, the last block's pooled rows take the small-call kernels
    ref = O.vit_forward(sd, cfg, O.preprocess_u8_exact_size(u8))
    tower = towers.VitTower(varch, sd, "cuda")
    assert tower._blocks[0].qkv_wf and tower._blocks[0].fc1_sf  # folded tensors are present
    tower.set_residual_stream(1) if hasattr(tower, "set_residual_stream") else None
    cos = lambda a, b: float((1 - torch.nn.functional.cosine_similarity(a.double().cpu(), b.double().cpu(), dim=-1)).max())
    try:
        _tune("ln_fold", 2)                          # statistics from the residual GEMMs' partial sums (default)
        folded = tower.encode_u8(u8.cuda())
        assert torch.equal(folded, tower.encode_u8(u8.cuda()))  # deterministic
        _tune("ln_fold", 1)                          # statistics from a read pass over the stream
        folded_rs = tower.encode_u8(u8.cuda())
        _tune("ln_fold", 0)
        plain = tower.encode_u8(u8.cuda())
    finally:
        _tune("ln_fold", 2)
    assert cos(folded, folded_rs) < 2e-5 and cos(folded_rs, ref) < 3e-4
    e_f, e_p, e_fp = cos(folded, ref), cos(plain, ref), cos(folded, plain)
    print(f"ViT-B/32 x4 layers ({tower.residual_stream=}): 1-cos vs oracle folded {e_f:.2e} plain {e_p:.2e}; folded vs plain {e_fp:.2e}")
    assert e_f < 3e-4 and e_p < 3e-4 and e_fp < 1e-4
    # text tower (causal, packed ragged sequences)
    tarch = archs.ClipTextArch(49408, 77, 512, 3, 8, 2048, 512)
    tcfg = O.ClipTextConfig(49408, 77, 512, 3, 8, 2048, 512)
    sdt = O.synthetic_clip_text_state_dict(tcfg, seed=9)
    ids = O.synthetic_clip_ids(40, seed=10)
    tt = towers.ClipTextTower(tarch, sdt, "cuda")
    reft = O.clip_text_forward(sdt, tcfg, ids)
    try:
        _tune("ln_fold", 2)
        f = tt.encode_ids(ids)
        _tune("ln_fold", 1)
        f1 = tt.encode_ids(ids)
        _tune("ln_fold", 0)
        p = tt.encode_ids(ids)
    finally:
        _tune("ln_fold", 2)
    assert cos(f, f1) < 2e-5
    assert cos(f, reft) < 3e-4 and cos(p, reft) < 3e-4 and cos(f, p) < 1e-4
