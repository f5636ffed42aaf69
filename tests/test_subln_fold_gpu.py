"""The EVA02 sub-LayerNorms folded into the GEMMs behind them (round 6, ABI 12): `mq_gemm_bf16_lnrs` (LN_APPLY + ROW_STATS in one launch: the out-projection /
fc2 form on the bf16 residual stream, and the gated (up | gate) form that leaves the product's row sums) and `mq_attention_stats` (the attention kernel's
per-head row sums), each against plain PyTorch fp32 and against the kernels they extend.
Reference arithmetic: timm eva.py EvaAttention (`norm` in front of `proj`) and SwiGLU (`norm` in front of `fc2`), reached through open_clip's TimmModel from
/root/reference/src/marqo/core/inference/embedding_models/open_clip_model.py:249-266 (registry rows model_registry.py:441-460)."""
import pytest
import torch

from marqo_amd import _lib as L

pytestmark = pytest.mark.gpu

GLU = 256


def _s():
    return torch.cuda.current_stream().cuda_stream


def _tune(**kw):
    lib = L.load()
    for k, v in kw.items():
        L.check(lib.mq_tune(k.encode(), v))


def _folded(W, b, gam, bet):
    wf = (W * gam.unsqueeze(0)).to(torch.bfloat16)
    return wf, (b + W @ bet).contiguous(), wf.float().sum(1).contiguous()


def _slot_sums(x, width):
    """(sum, sum of squares) of the rows of x [M, N] per `width`-column slot, fp64"""
    M, N = x.shape
    ns = (N + width - 1) // width
    pad = torch.zeros(M, ns * width, device=x.device, dtype=torch.float64)
    pad[:, :N] = x.double()
    pad = pad.view(M, ns, width)
    return pad.sum(-1), pad.pow(2).sum(-1)


@pytest.mark.parametrize("plan", [dict(), dict(gemm_mt=2), dict(gemm_mt=6), dict(gemm_nh=3)])
@pytest.mark.parametrize("M,N,K", [(12608, 768, 2048), (4099, 768, 768), (1000, 1024, 2752), (333, 512, 64), (130, 72, 128)])
def test_residual_gemm_applies_the_layernorm_of_its_input_and_leaves_row_sums(M, N, K, plan):
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = (torch.randn(M, K, device="cuda", generator=g) * 1.7 + 0.3).to(torch.bfloat16)       # the rows the LayerNorm normalises (attention output / gated product)
    a[:, 3] += 9.0
    gam, bet = 1 + 0.2 * torch.randn(K, device="cuda", generator=g), 0.1 * torch.randn(K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = 0.1 * torch.randn(N, device="cuda", generator=g)
    x0 = (torch.randn(M, N, device="cuda", generator=g) * 2 + 0.5).to(torch.bfloat16)        # the residual stream
    eps = 1e-6
    want = x0.float() + torch.nn.functional.layer_norm(a.float(), (K,), gam, bet, eps) @ W.t() + b
    wf, bf, colsum = _folded(W, b, gam, bet)
    stats = torch.empty(M, 2, device="cuda")
    ad = a.double()
    stats[:, 0] = ad.mean(1).float()
    stats[:, 1] = (1.0 / torch.sqrt(ad.var(1, unbiased=False) + eps)).float()
    ns = (N + 63) // 64
    flags = L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL
    try:
        _tune(**plan)
        x = x0.clone()
        part = torch.full((ns, M, 2), float("nan"), device="cuda")       # slot-major
        L.check(lib.mq_gemm_bf16_lnrs(a.data_ptr(), K, wf.data_ptr(), K, bf.data_ptr(), colsum.data_ptr(), stats.data_ptr(), x.data_ptr(), x.data_ptr(), N, M, N, K, flags, part.data_ptr(), _s()))
        scale = want.abs().max().item()
        assert (x.float() - want).abs().max().item() / scale < 2.5e-2
        assert ((x.float() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item() < 6e-3
        # the un-folded engine path for scale: LayerNorm -> bf16 -> residual GEMM
        hn = torch.nn.functional.layer_norm(a.float(), (K,), gam, bet, eps).to(torch.bfloat16)
        unf = x0.clone()
        Wb = W.to(torch.bfloat16)
        L.check(lib.mq_gemm_bf16(hn.data_ptr(), K, Wb.data_ptr(), K, b.data_ptr(), unf.data_ptr(), unf.data_ptr(), N, M, N, K, flags, _s()))
        rms = lambda t: ((t.float() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
        assert rms(x) < 2.0 * rms(unf) + 5e-4
        # the row sums are those of the stored (rounded) rows, every (row, slot) written
        assert not torch.isnan(part).any()
        s1, s2 = _slot_sums(x, 64)
        assert torch.allclose(part[..., 0].double().t(), s1, rtol=1e-5, atol=1e-3) and torch.allclose(part[..., 1].double().t(), s2, rtol=1e-5, atol=1e-3)
        # deterministic, and the tile plan only changes the schedule
        x2, part2 = x0.clone(), torch.empty_like(part)
        L.check(lib.mq_gemm_bf16_lnrs(a.data_ptr(), K, wf.data_ptr(), K, bf.data_ptr(), colsum.data_ptr(), stats.data_ptr(), x2.data_ptr(), x2.data_ptr(), N, M, N, K, flags, part2.data_ptr(), _s()))
        assert torch.equal(x2, x) and torch.equal(part2, part)
        _tune(gemm_mt=0, gemm_nh=0)
        x3, part3 = x0.clone(), torch.empty_like(part)
        L.check(lib.mq_gemm_bf16_lnrs(a.data_ptr(), K, wf.data_ptr(), K, bf.data_ptr(), colsum.data_ptr(), stats.data_ptr(), x3.data_ptr(), x3.data_ptr(), N, M, N, K, flags, part3.data_ptr(), _s()))
        assert torch.equal(x3, x) and torch.equal(part3, part)
    finally:
        _tune(gemm_mt=0, gemm_nh=0)


@pytest.mark.parametrize("plan", [dict(), dict(gemm_mt=2), dict(gemm_mt=6), dict(gemm_nh=3)])
@pytest.mark.parametrize("M,F,K", [(12608, 2048, 768), (4099, 192, 128), (1000, 2752, 1024), (300, 64, 64)])
def test_gated_gemm_leaves_the_row_sums_of_the_product(M, F, K, plan):
    """the (up | gate) GEMM with the LayerNorm in front folded in (LN_APPLY) and MQ_EPI_GLU: the same bits as mq_gemm_bf16_ln's gated form, plus the product's
    (sum, sum of squares) per row and 32-unit slot"""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(M + F)
    x = (torch.randn(M, K, device="cuda", generator=g) * 1.5 + 0.2).to(torch.bfloat16)
    gam, bet = 1 + 0.2 * torch.randn(K, device="cuda", generator=g), 0.1 * torch.randn(K, device="cuda", generator=g)
    Wu, Wg = torch.randn(F, K, device="cuda", generator=g) / K ** 0.5, torch.randn(F, K, device="cuda", generator=g) / K ** 0.5
    bu, bg = torch.randn(F, device="cuda", generator=g), torch.randn(F, device="cuda", generator=g)
    il = lambda u, v: torch.stack([u.reshape(F // 16, 16, *u.shape[1:]), v.reshape(F // 16, 16, *v.shape[1:])], dim=1).reshape(2 * F, *u.shape[1:]).contiguous()
    wf, bf, colsum = _folded(il(Wu, Wg), il(bu, bg), gam, bet)
    eps = 1e-6
    stats = torch.empty(M, 2, device="cuda")
    L.check(lib.mq_row_stats(x.data_ptr(), stats.data_ptr(), M, K, eps, _s()))
    ldc, N = 2 * F, 2 * F
    ns = (N + 63) // 64
    try:
        _tune(**plan)
        ref = torch.full((M, ldc), 7.0, device="cuda", dtype=torch.bfloat16)
        L.check(lib.mq_gemm_bf16_ln(x.data_ptr(), K, wf.data_ptr(), K, bf.data_ptr(), colsum.data_ptr(), stats.data_ptr(), ref.data_ptr(), ldc, M, N, K, L.MQ_EPI_BIAS | GLU, _s()))
        out = torch.full((M, ldc), 7.0, device="cuda", dtype=torch.bfloat16)
        part = torch.full((ns, M, 2), float("nan"), device="cuda")       # slot-major
        L.check(lib.mq_gemm_bf16_lnrs(x.data_ptr(), K, wf.data_ptr(), K, bf.data_ptr(), colsum.data_ptr(), stats.data_ptr(), 0, out.data_ptr(), ldc, M, N, K, L.MQ_EPI_BIAS | GLU, part.data_ptr(), _s()))
        assert torch.equal(out, ref)
        hn = torch.nn.functional.layer_norm(x.float(), (K,), gam, bet, eps)
        want = (hn @ Wu.t() + bu) * torch.nn.functional.silu(hn @ Wg.t() + bg)
        assert (out[:, :F].float() - want).abs().max().item() / (want.abs().max().item() + 1e-6) < 2.5e-2
        assert not torch.isnan(part).any()
        s1, s2 = _slot_sums(out[:, :F], 32)
        assert torch.allclose(part[..., 0].double().t(), s1, rtol=1e-5, atol=1e-3) and torch.allclose(part[..., 1].double().t(), s2, rtol=1e-5, atol=1e-3)
        # ... which the finalise kernel turns into the statistics of the LayerNorm over the product's first `F` columns
        fin = torch.empty(M, 2, device="cuda")
        L.check(lib.mq_row_stats_finalize(part.data_ptr(), ns, fin.data_ptr(), M, F, eps, _s()))
        pd = out[:, :F].double()
        assert torch.allclose(fin[:, 0].double(), pd.mean(1), rtol=1e-4, atol=1e-5)
        assert torch.allclose(fin[:, 1].double(), 1.0 / torch.sqrt(pd.var(1, unbiased=False) + eps), rtol=2e-4)
    finally:
        _tune(gemm_mt=0, gemm_nh=0)


def test_attention_leaves_the_row_sums_per_head():
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(11)
    for W, heads, nseq, fixed, lens, mask in ((768, 12, 37, 197, None, 0), (1024, 16, 5, 257, None, 0), (512, 8, 6, 0, [3, 77, 1, 40, 64, 65], 1), (384, 3, 9, 50, None, 0)):
        if lens is None:
            rows, cu, mx = nseq * fixed, 0, fixed
        else:
            cu_t = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32, device="cuda")
            rows, cu, mx = int(cu_t[-1]), cu_t.data_ptr(), max(lens)
        qkv = torch.randn(rows, 3 * W, device="cuda", generator=g).to(torch.bfloat16)
        want = torch.empty(rows, W, device="cuda", dtype=torch.bfloat16)
        L.check(lib.mq_attention(qkv.data_ptr(), want.data_ptr(), cu, nseq, fixed, mx, W, heads, mask, _s()))
        got = torch.empty_like(want)
        part = torch.full((heads, rows, 2), float("nan"), device="cuda")    # slot-major
        L.check(lib.mq_attention_stats(qkv.data_ptr(), got.data_ptr(), cu, nseq, fixed, mx, W, heads, mask, part.data_ptr(), rows, _s()))
        assert torch.equal(got, want) and not torch.isnan(part).any()
        s1, s2 = _slot_sums(got, W // heads)
        assert torch.allclose(part[..., 0].double().t(), s1, rtol=1e-5, atol=1e-4) and torch.allclose(part[..., 1].double().t(), s2, rtol=1e-5, atol=1e-4)
        again = torch.empty_like(part)
        L.check(lib.mq_attention_stats(qkv.data_ptr(), got.data_ptr(), cu, nseq, fixed, mx, W, heads, mask, again.data_ptr(), rows, _s()))
        assert torch.equal(again, part)
        fin = torch.empty(rows, 2, device="cuda")
        L.check(lib.mq_row_stats_finalize(part.data_ptr(), heads, fin.data_ptr(), rows, W, 1e-6, _s()))
        gd = got.double()
        assert torch.allclose(fin[:, 0].double(), gd.mean(1), rtol=1e-4, atol=1e-5)
        assert torch.allclose(fin[:, 1].double(), 1.0 / torch.sqrt(gd.var(1, unbiased=False) + 1e-6), rtol=2e-4)
