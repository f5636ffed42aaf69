"""Parity of the individual HIP kernels (through the C ABI) against plain PyTorch fp32 on the
same inputs.  These are floating-point kernels: bf16 operands, fp32 accumulation; tolerances are
stated per test."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from marqo_amd import _lib
    return _lib.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _gemm(lib, A, W, bias=None, residual=None, flags=0):
    from marqo_amd import _lib as L
    M, K = A.shape
    N = W.shape[0]
    out_f32 = bool(flags & L.MQ_EPI_OUT_F32)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if out_f32 else torch.bfloat16)
    if residual is not None:
        out.copy_(residual)
        residual = out
    L.check(lib.mq_gemm_bf16(A.data_ptr(), K, W.data_ptr(), K, L.ptr(bias), L.ptr(residual), out.data_ptr(), N,
                             M, N, K, flags, _stream()), "gemm")
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 768, 768), (50 * 7, 2304, 768), (1000, 512, 3072),
                                   (12800, 768, 768), (37, 512, 768), (1, 768, 1024), (257 * 3, 4096, 1024),
                                   (300, 132, 128)])
def test_gemm_plain_matches_fp32(lib, M, N, K):
    from marqo_amd import _lib as L
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    # asymmetric, non-random structure added so a transposed / permuted write cannot pass
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05
         + torch.arange(N, device="cuda")[:, None] * 1e-3).to(torch.bfloat16)
    ref = A.float() @ W.float().t()
    out = _gemm(lib, A, W, flags=L.MQ_EPI_OUT_F32)
    # bf16 inputs are exact in both; only fp32 accumulation order differs
    assert torch.allclose(out, ref, rtol=1e-4, atol=1e-3 * math.sqrt(K / 64)), (out - ref).abs().max()
    out_b = _gemm(lib, A, W, flags=0)
    assert torch.allclose(out_b.float(), ref, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("act", ["gelu", "quickgelu"])
def test_gemm_bias_act(lib, act):
    from marqo_amd import _lib as L
    g = torch.Generator(device="cuda").manual_seed(2)
    M, N, K = 333, 3072, 768
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g)
    pre = A.float() @ W.float().t() + b
    if act == "gelu":
        ref = torch.nn.functional.gelu(pre)
        flag = L.MQ_EPI_GELU
    else:
        ref = pre * torch.sigmoid(1.702 * pre)
        flag = L.MQ_EPI_QUICKGELU
    out = _gemm(lib, A, W, bias=b, flags=L.MQ_EPI_BIAS | flag)
    assert torch.allclose(out.float(), ref, rtol=1e-2, atol=1e-2), (out.float() - ref).abs().max()


def test_gemm_bias_residual_inplace(lib):
    from marqo_amd import _lib as L
    g = torch.Generator(device="cuda").manual_seed(3)
    M, N, K = 777, 768, 3072
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g)
    ref = A.float() @ W.float().t() + b + res
    out = _gemm(lib, A, W, bias=b, residual=res, flags=L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32)
    assert torch.allclose(out, ref, rtol=1e-4, atol=5e-3), (out - ref).abs().max()


def test_gemm_rejects_bad_shapes(lib):
    from marqo_amd import _lib as L
    A = torch.zeros(8, 96, device="cuda", dtype=torch.bfloat16)
    W = torch.zeros(8, 96, device="cuda", dtype=torch.bfloat16)
    out = torch.zeros(8, 8, device="cuda")
    rc = lib.mq_gemm_bf16(A.data_ptr(), 96, W.data_ptr(), 96, 0, 0, out.data_ptr(), 8, 8, 8, 96, L.MQ_EPI_OUT_F32, _stream())
    assert rc != 0 and b"multiple of 64" in lib.mq_last_error()


@pytest.mark.parametrize("rows,W", [(5, 768), (1001, 1024), (64, 512), (3, 1280), (7, 384), (9, 1408), (130, 1664), (6, 2048)])
def test_layernorm(lib, rows, W):
    from marqo_amd import _lib as L
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn(rows, W, device="cuda", generator=g) * 3 + 1.5
    gam = torch.randn(W, device="cuda", generator=g)
    bet = torch.randn(W, device="cuda", generator=g)
    ref = torch.nn.functional.layer_norm(x, (W,), gam, bet, 1e-5)
    ob = torch.empty(rows, W, device="cuda", dtype=torch.bfloat16)
    of = torch.empty(rows, W, device="cuda")
    L.check(lib.mq_layernorm(x.data_ptr(), 0, gam.data_ptr(), bet.data_ptr(), ob.data_ptr(), of.data_ptr(), rows, W, 1e-5, _stream()))
    torch.cuda.synchronize()
    assert torch.allclose(of, ref, rtol=1e-5, atol=1e-5)
    assert torch.allclose(ob.float(), ref, rtol=1e-2, atol=1e-2)
    # gathered rows
    idx = torch.tensor([rows - 1, 0, rows // 2], device="cuda", dtype=torch.int32)
    og = torch.empty(3, W, device="cuda")
    L.check(lib.mq_layernorm(x.data_ptr(), idx.data_ptr(), gam.data_ptr(), bet.data_ptr(), 0, og.data_ptr(), 3, W, 1e-5, _stream()))
    torch.cuda.synchronize()
    assert torch.allclose(og, ref[idx.long()], rtol=1e-5, atol=1e-5)


def _ref_attention(qkv, lens, heads, causal, hd=64):
    W = qkv.shape[1] // 3
    out = torch.zeros(qkv.shape[0], W, device=qkv.device)
    r0 = 0
    for ln in lens:
        blk = qkv[r0:r0 + ln].float()
        q, k, v = blk[:, :W], blk[:, W:2 * W], blk[:, 2 * W:]
        q = q.view(ln, heads, hd).transpose(0, 1)
        k = k.view(ln, heads, hd).transpose(0, 1)
        v = v.view(ln, heads, hd).transpose(0, 1)
        s = q @ k.transpose(1, 2) / hd ** 0.5
        if causal:
            s = s + torch.full((ln, ln), float("-inf"), device=qkv.device).triu(1)
        p = torch.softmax(s, dim=-1)
        out[r0:r0 + ln] = (p @ v).transpose(0, 1).reshape(ln, W)
        r0 += ln
    return out


@pytest.mark.parametrize("lens,heads,causal,hd", [([50] * 6, 12, False, 64), ([257] * 3, 16, False, 64), ([77] * 5, 8, True, 64),
                                                    ([5, 77, 1, 33, 64, 65], 12, True, 64), ([9, 512, 17, 128], 12, False, 64),
                                                    ([16], 2, False, 64),
                                                    # 65..80 tokens: five query blocks (the opt-in five-wave workgroups are checked bit for bit below)
                                                    ([77, 77, 66, 80, 3, 77], 12, False, 64), ([80] * 3, 4, True, 64),
                                                    # 128-wide heads (ViT-H / g / bigG after padding): 257 tokens fill the 160 KiB of LDS
                                                    ([257] * 3, 16, False, 128), ([5, 77, 1, 33, 64, 65, 320], 3, True, 128),
                                                    ([50] * 4, 5, False, 128), ([16], 1, False, 128),
                                                    # head strides 96 / 112 (80 / 88 / 104-wide heads padded): LDS rows stay 256 B
                                                    ([257] * 3, 16, False, 96), ([257] * 2, 16, False, 112), ([5, 77, 1, 33, 64, 65, 320], 3, True, 96),
                                                    ([5, 77, 1, 33, 64, 65, 320], 3, True, 112), ([50] * 4, 5, False, 96), ([16], 1, False, 112),
                                                    # longer than the LDS holds (640 keys at 64-wide, 320 at wider heads): K / V stream through it
                                                    ([730, 1000, 65], 2, False, 96), ([729] * 2, 3, False, 128), ([1024, 700], 2, False, 64),
                                                    ([900, 100, 641], 2, True, 64), ([700, 321], 1, True, 112)])
def test_attention(lib, lens, heads, causal, hd):
    from marqo_amd import _lib as L
    g = torch.Generator(device="cuda").manual_seed(5)
    W = heads * hd
    rows = sum(lens)
    qkv = torch.randn(rows, 3 * W, device="cuda", generator=g).to(torch.bfloat16)
    ref = _ref_attention(qkv, lens, heads, causal, hd)
    out = torch.empty(rows, W, device="cuda", dtype=torch.bfloat16)
    fixed = lens[0] if len(set(lens)) == 1 else 0
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), device="cuda", dtype=torch.int32)
    L.check(lib.mq_attention(qkv.data_ptr(), out.data_ptr(), 0 if fixed else cu.data_ptr(), len(lens), fixed, max(lens), W, heads,
                             L.MQ_MASK_CAUSAL if causal else L.MQ_MASK_NONE, _stream()))
    torch.cuda.synchronize()
    err = (out.float() - ref).abs().max().item()
    # P is rounded to bf16 before P.V (8 mantissa bits) and the output is bf16
    assert err < 2e-2, err
    # 4 or 8 waves per workgroup (auto: 8 once K + V exceed 80 KiB of LDS) walk the same query blocks: identical bits
    try:
        for nw in (4, 8, 5):
            L.check(lib.mq_tune(b"attn_waves", nw))
            out2 = torch.empty_like(out)
            L.check(lib.mq_attention(qkv.data_ptr(), out2.data_ptr(), 0 if fixed else cu.data_ptr(), len(lens), fixed, max(lens), W, heads,
                                     L.MQ_MASK_CAUSAL if causal else L.MQ_MASK_NONE, _stream()))
            torch.cuda.synchronize()
            assert torch.equal(out2, out), nw
    finally:
        L.check(lib.mq_tune(b"attn_waves", 0))


@pytest.mark.parametrize("lens,heads", [([50] * 4, 12), ([5, 77, 1, 33, 64, 65], 12), ([200, 129, 17], 2), ([512], 3)])
def test_attention_with_relative_position_bias(lib, lens, heads):
    """mq_attention_bias: scores / sqrt(d) + bias[h][key - query] before the softmax (MPNet), the table pre-multiplied by sqrt(d)"""
    from marqo_amd import _lib as L
    g = torch.Generator(device="cuda").manual_seed(7)
    hd, W, rows, span = 64, heads * 64, sum(lens), 512
    qkv = torch.randn(rows, 3 * W, device="cuda", generator=g).to(torch.bfloat16)
    bias = torch.randn(heads, 2 * span - 1, device="cuda", generator=g) * 2.0
    ref = torch.empty(rows, W, device="cuda")
    r0 = 0
    for ln in lens:
        blk = qkv[r0:r0 + ln].float()
        q, k, v = [t.reshape(ln, heads, hd).transpose(0, 1) for t in blk.split(W, dim=1)]
        pos = torch.arange(ln, device="cuda")
        b = bias[:, (pos[None, :] - pos[:, None]) + span - 1]          # [H, q, k]
        p = torch.softmax(q @ k.transpose(1, 2) / hd ** 0.5 + b, dim=-1)
        ref[r0:r0 + ln] = (p @ v).transpose(0, 1).reshape(ln, W)
        r0 += ln
    out = torch.empty(rows, W, device="cuda", dtype=torch.bfloat16)
    fixed = lens[0] if len(set(lens)) == 1 else 0
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), device="cuda", dtype=torch.int32)
    table = (bias * hd ** 0.5).contiguous()
    L.check(lib.mq_attention_bias(qkv.data_ptr(), out.data_ptr(), 0 if fixed else cu.data_ptr(), len(lens), fixed, max(lens), W, heads,
                                  table.data_ptr(), span, _stream()))
    torch.cuda.synchronize()
    assert (out.float() - ref).abs().max().item() < 2e-2
    plain = torch.empty_like(out)
    L.check(lib.mq_attention(qkv.data_ptr(), plain.data_ptr(), 0 if fixed else cu.data_ptr(), len(lens), fixed, max(lens), W, heads,
                             L.MQ_MASK_NONE, _stream()))
    zero = torch.zeros_like(table)
    L.check(lib.mq_attention_bias(qkv.data_ptr(), out.data_ptr(), 0 if fixed else cu.data_ptr(), len(lens), fixed, max(lens), W, heads,
                                  zero.data_ptr(), span, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(out, plain)                                   # a zero table is exactly the plain kernel
    assert lib.mq_attention_bias(qkv.data_ptr(), out.data_ptr(), 0 if fixed else cu.data_ptr(), len(lens), fixed, max(lens), W, heads,
                                 table.data_ptr(), max(lens) - 1, _stream()) == -1   # rel_span must cover the longest sequence

