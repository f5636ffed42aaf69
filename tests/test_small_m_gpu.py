"""The search path: one query (or a handful of rows) per call goes through the column-sliced skinny GEMMs of csrc/gemm_small.hip, with the
LayerNorm of pre-LN blocks fused into the GEMM prologue.  Checked here: the two kernels against a fp32 reference on the bf16-rounded operands
(every epilogue, ragged M / N, both residual stream types); the fused LayerNorm bit for bit against mq_layernorm + the same GEMM; whole towers
on the skinny path against the CPU oracle and against the tiled kernels; row independence (an embedding does not depend on what shares its call)."""
import pytest
import torch

from marqo_amd import _lib as L
from oracle import towers as O

pytestmark = pytest.mark.gpu


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _small(lib, A, W, bias, res, flags, out_dtype):
    M, K = A.shape
    N = W.shape[0]
    out = res.clone() if res is not None else torch.empty(M, N, device="cuda", dtype=out_dtype)
    L.check(lib.mq_gemm_small_bf16(A.data_ptr(), K, W.data_ptr(), K, L.ptr(bias), out.data_ptr() if res is not None else 0, out.data_ptr(), N,
                                   M, N, K, flags, _stream()), "mq_gemm_small_bf16")
    return out


@pytest.mark.parametrize("M", [1, 7, 16, 17, 33, 50, 64, 80])
def test_skinny_gemm_matches_reference(M):
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(M)
    shapes = ((512, 512), (2304, 768), (768, 3072), (100, 96), (4, 32), (1024, 4096), (3072, 1024), (1288, 64))
    for (N, K) in shapes:
        A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda", generator=g)
        res = torch.randn(M, N, device="cuda", generator=g)
        ref = A.float() @ W.float().t()
        B, G, Q, R, F = L.MQ_EPI_BIAS, L.MQ_EPI_GELU, L.MQ_EPI_QUICKGELU, L.MQ_EPI_RESIDUAL, L.MQ_EPI_OUT_F32
        rb = ref + bias
        cases = ((0, None, ref), (F, None, ref), (B, None, rb), (B | G, None, torch.nn.functional.gelu(rb)),
                 (B | Q, None, rb * torch.sigmoid(1.702 * rb)), (B | R | F, res, rb + res),
                 (B | R, res.to(torch.bfloat16), rb + res.to(torch.bfloat16).float()))
        for flags, r, want in cases:
            f32 = bool(flags & F)
            out = _small(lib, A, W, bias, r, flags, torch.float32 if f32 else torch.bfloat16).float()
            err = (out - want).abs().max().item() / (want.abs().max().item() + 1e-6)
            assert err < (2e-3 if f32 else 2e-2), (M, N, K, flags, err)
            again = _small(lib, A, W, bias, r, flags, torch.float32 if f32 else torch.bfloat16).float()
            assert torch.equal(out, again)                         # fixed summation order
        # row independence: row m of an M-row call == the same row in a call of its own
        one = _small(lib, A[M // 2:M // 2 + 1].contiguous(), W, bias, None, B, torch.bfloat16)
        assert torch.equal(one[0], _small(lib, A, W, bias, None, B, torch.bfloat16)[M // 2])
    # the public GEMM routes small M here by itself: identical bits
    A = torch.randn(M, 768, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(512, 768, device="cuda", generator=g) / 28).to(torch.bfloat16)
    bias = torch.randn(512, device="cuda", generator=g)
    via = torch.empty(M, 512, device="cuda", dtype=torch.bfloat16)
    L.check(lib.mq_gemm_bf16(A.data_ptr(), 768, W.data_ptr(), 768, L.ptr(bias), 0, via.data_ptr(), 512, M, 512, 768, L.MQ_EPI_BIAS, _stream()))
    assert torch.equal(via, _small(lib, A, W, bias, None, L.MQ_EPI_BIAS, torch.bfloat16))


@pytest.mark.parametrize("M", [1, 10, 16, 17, 32])
def test_fused_layernorm_gemm(M):
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(100 + M)
    for (N, K) in ((1536, 512), (2304, 768), (3072, 768), (4096, 1024), (5120, 1280), (100, 96)):
        for xb in (0, 1):
            x = torch.randn(M, K, device="cuda", generator=g) * 3 + 0.5
            if xb:
                x = x.to(torch.bfloat16)
            gam = torch.rand(K, device="cuda", generator=g) + 0.5
            bet = torch.randn(K, device="cuda", generator=g) * 0.1
            W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
            bias = torch.randn(N, device="cuda", generator=g)
            h = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
            hf = torch.empty(M, K, device="cuda")
            L.check(lib.mq_layernorm_ex(x.data_ptr(), xb, 0, gam.data_ptr(), bet.data_ptr(), h.data_ptr(), hf.data_ptr(), M, K, 1e-5, _stream()))
            for flags in (L.MQ_EPI_BIAS, L.MQ_EPI_BIAS | L.MQ_EPI_GELU, L.MQ_EPI_BIAS | L.MQ_EPI_QUICKGELU):
                out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
                xn = torch.full((M, K), float("nan"), device="cuda")
                L.check(lib.mq_ln_gemm_small_bf16(x.data_ptr(), K, xb, gam.data_ptr(), bet.data_ptr(), 1e-5, W.data_ptr(), K, bias.data_ptr(),
                                                  out.data_ptr(), N, M, N, K, flags, xn.data_ptr(), _stream()), "mq_ln_gemm_small_bf16")
                assert torch.equal(xn, hf)                                             # the fp32 normalised rows (post-LN residual)
                two = _small(lib, h, W, bias, None, flags, torch.bfloat16)
                assert torch.equal(out, two), (M, N, K, xb, flags, (out.float() - two.float()).abs().max().item())
                ref = torch.nn.functional.layer_norm(x.float(), (K,), gam, bet, 1e-5).to(torch.bfloat16).float() @ W.float().t() + bias
                if flags & L.MQ_EPI_GELU:
                    ref = torch.nn.functional.gelu(ref)
                if flags & L.MQ_EPI_QUICKGELU:
                    ref = ref * torch.sigmoid(1.702 * ref)
                err = (out.float() - ref).abs().max().item() / (ref.abs().max().item() + 1e-6)
                assert err < 3e-2, (M, N, K, xb, flags, err)
    # shapes outside the fused kernel's range are refused, not mis-run
    x = torch.zeros(33, 768, device="cuda")
    assert lib.mq_ln_gemm_small_bf16(x.data_ptr(), 768, 0, x.data_ptr(), x.data_ptr(), 1e-5, x.data_ptr(), 768, x.data_ptr(), x.data_ptr(), 768,
                                     33, 768, 768, L.MQ_EPI_BIAS, 0, _stream()) != 0
    y = torch.zeros(8, 768, device="cuda")   # in-place normalisation would race with the other workgroups' reads
    assert lib.mq_ln_gemm_small_bf16(y.data_ptr(), 768, 0, y.data_ptr(), y.data_ptr(), 1e-5, y.data_ptr(), 768, y.data_ptr(), x.data_ptr(), 768,
                                     8, 768, 768, L.MQ_EPI_BIAS, y.data_ptr(), _stream()) != 0


def _cos_err(a, b):
    a, b = a.double().cpu(), torch.as_tensor(b).double().cpu()
    return float((1 - torch.nn.functional.cosine_similarity(a, b, dim=-1)).max())


def test_single_query_towers_on_the_skinny_path():
    """One query per call, full-size towers (ViT-B/32 text + image, e5-base): skinny path vs the fp32 CPU oracle, vs the tiled kernels on
    the same input, and bit-identical to the same query inside a small batch."""
    from marqo_amd.engine import archs, synthetic, towers
    lib = L.load()
    v, t = archs.resolve_open_clip("ViT-B-32")
    sd = synthetic.random_open_clip_state_dict(vision=v, text=t, seed=0)
    tt, vt = towers.ClipTextTower(t, sd, "cuda"), towers.VitTower(v, sd, "cuda")
    ids = torch.zeros(5, t.ctx, dtype=torch.int64)                                     # SOT, L tokens, EOT (the max id), zero pad
    for i, n_tok in enumerate((8, 12, 20, 30, 3)):
        ids[i, 0], ids[i, 1 + n_tok] = t.vocab - 2, t.vocab - 1
        ids[i, 1:1 + n_tok] = torch.randint(1, t.vocab - 2, (n_tok,), generator=torch.Generator().manual_seed(i))
    tcfg = O.ClipTextConfig(t.vocab, t.ctx, t.width, t.layers, t.heads, t.mlp_dim, t.out_dim)
    ref_t = O.clip_text_forward(sd, tcfg, ids)
    u8 = O.synthetic_images_u8(2, v.image_size, seed=4)
    vcfg = O.VitConfig(v.image_size, v.patch_size, v.width, v.layers, v.heads, v.mlp_dim, v.out_dim)
    ref_v = O.vit_forward(sd, vcfg, O.preprocess_u8_exact_size(u8))
    b = archs.HF_BERT_ARCHS["intfloat/e5-base-v2"]
    bsd = synthetic.random_bert_state_dict(b, seed=0)
    bt = towers.BertTower(b, bsd, "cuda")
    bids = torch.randint(1000, b.vocab, (3, 12), generator=torch.Generator().manual_seed(5))
    bmask = torch.ones(3, 12, dtype=torch.int64)
    bcfg = O.BertConfig(vocab=b.vocab, max_pos=b.max_pos, width=b.width, layers=b.layers, heads=b.heads, mlp_dim=b.mlp_dim, ln_eps=b.ln_eps)
    ref_b = O.hf_encode(bsd, bcfg, bids, bmask)
    try:
        skinny_t = torch.cat([tt.encode_ids(ids[i:i + 1]) for i in range(5)])        # graph replay of the skinny path, one query each
        skinny_v = torch.cat([vt.encode_u8(u8[i:i + 1].cuda()) for i in range(2)])   # 50 rows
        skinny_b = torch.cat([bt.encode_ids(bids[i:i + 1], bmask[i:i + 1]) for i in range(3)])
        assert _cos_err(skinny_t, ref_t) < 1e-4 and _cos_err(skinny_v, ref_v) < 1e-4 and _cos_err(skinny_b, ref_b) < 1e-4
        print(f"single-query skinny path: 1-cos vs fp32 oracle  text {_cos_err(skinny_t, ref_t):.2e}  image {_cos_err(skinny_v, ref_v):.2e}  "
              f"e5 {_cos_err(skinny_b, ref_b):.2e}")
        # a query inside a small batch (still <= 80 rows): the same bits as on its own — rows are independent in the skinny kernels, and the
        # fused LayerNorm (<= 32 rows) has the arithmetic of the stand-alone one
        short = ids[:3]
        assert torch.equal(tt.encode_ids(short), skinny_t[:3])
        assert torch.equal(bt.encode_ids(bids, bmask), skinny_b)
        L.check(lib.mq_tune(b"small_m", 0))
        for tower in (tt, vt, bt):
            tower._graphs.clear()                                                     # the captured single-request graphs hold skinny launches
        tiled_t = torch.cat([tt.encode_ids(ids[i:i + 1]) for i in range(5)])
        tiled_v = torch.cat([vt.encode_u8(u8[i:i + 1].cuda()) for i in range(2)])
        tiled_b = torch.cat([bt.encode_ids(bids[i:i + 1], bmask[i:i + 1]) for i in range(3)])
        for a, c in ((skinny_t, tiled_t), (skinny_v, tiled_v), (skinny_b, tiled_b)):
            assert _cos_err(a, c) < 1e-4   # two summation orders of the same products (and, under the bf16 residual stream, of the same bf16-rounded x)
    finally:
        L.check(lib.mq_tune(b"small_m", 80))


@pytest.mark.parametrize("M", [81, 100, 192, 256, 320])
def test_skinny_gemm_in_row_groups(M):
    """81..320 rows (the pooled rows of a 256-item batch's last block; a request of a few items): the same kernel, one workgroup per
    (16-column slice, <= 80-row group).  Against the fp32 reference for every epilogue, deterministic, and every row bit-identical to the
    same row in a ONE-group call of its own (a row's arithmetic does not depend on its group), which is also what mq_gemm_bf16 routes."""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(1000 + M)
    B, G, Q, R, F = L.MQ_EPI_BIAS, L.MQ_EPI_GELU, L.MQ_EPI_QUICKGELU, L.MQ_EPI_RESIDUAL, L.MQ_EPI_OUT_F32
    for (N, K) in ((768, 768), (768, 3072), (3072, 768), (2304, 768), (512, 768), (100, 96), (1024, 4096)):
        A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda", generator=g)
        res = torch.randn(M, N, device="cuda", generator=g)
        ref = A.float() @ W.float().t()
        rb = ref + bias
        cases = ((0, None, ref), (F, None, ref), (B, None, rb), (B | G, None, torch.nn.functional.gelu(rb)),
                 (B | Q, None, rb * torch.sigmoid(1.702 * rb)), (B | R | F, res, rb + res),
                 (B | R, res.to(torch.bfloat16), rb + res.to(torch.bfloat16).float()))
        for flags, r, want in cases:
            f32 = bool(flags & F)
            dt = torch.float32 if f32 else torch.bfloat16
            out = _small(lib, A, W, bias, r, flags, dt)
            err = (out.float() - want).abs().max().item() / (want.abs().max().item() + 1e-6)
            assert err < (2e-3 if f32 else 2e-2), (M, N, K, flags, err)
            assert torch.equal(out, _small(lib, A, W, bias, r, flags, dt))
            for m in (0, 79, 80, M - 1):                       # first group, both sides of a group edge, the ragged tail
                one = _small(lib, A[m:m + 1].contiguous(), W, bias, None if r is None else r[m:m + 1].contiguous(), flags, dt)
                assert torch.equal(one[0], out[m]), (M, N, K, flags, m)
        if K % 64 == 0:                                            # (the public GEMM's own shape rule)
            via = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            L.check(lib.mq_gemm_bf16(A.data_ptr(), K, W.data_ptr(), K, L.ptr(bias), 0, via.data_ptr(), N, M, N, K, B, _stream()))
            assert torch.equal(via, _small(lib, A, W, bias, None, B, torch.bfloat16))
    # past the knob the public GEMM is the tiled kernel again, and the skinny entry point refuses
    A = torch.randn(321, 768, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(512, 768, device="cuda", generator=g) / 28).to(torch.bfloat16)
    out = torch.empty(321, 512, device="cuda", dtype=torch.bfloat16)
    assert lib.mq_gemm_small_bf16(A.data_ptr(), 768, W.data_ptr(), 768, 0, 0, out.data_ptr(), 512, 321, 512, 768, 0, _stream()) != 0
    try:
        L.check(lib.mq_tune(b"small_m_grouped", 0))
        assert lib.mq_gemm_small_bf16(A.data_ptr(), 768, W.data_ptr(), 768, 0, 0, out.data_ptr(), 512, 256, 512, 768, 0, _stream()) != 0
    finally:
        L.check(lib.mq_tune(b"small_m_grouped", 320))


def test_few_item_requests_on_the_grouped_path():
    """4 images (200 rows) / 16 short texts: every block's GEMMs run in row groups.  Against the fp32 oracle and against the tiled kernels."""
    from marqo_amd.engine import archs, synthetic, towers
    lib = L.load()
    v, t = archs.resolve_open_clip("ViT-B-32")
    sd = synthetic.random_open_clip_state_dict(vision=v, text=t, seed=0)
    tt, vt = towers.ClipTextTower(t, sd, "cuda"), towers.VitTower(v, sd, "cuda")
    ids = torch.zeros(16, t.ctx, dtype=torch.int64)
    for i in range(16):
        n_tok = 6 + i
        ids[i, 0], ids[i, 1 + n_tok] = t.vocab - 2, t.vocab - 1
        ids[i, 1:1 + n_tok] = torch.randint(1, t.vocab - 2, (n_tok,), generator=torch.Generator().manual_seed(i))
    u8 = O.synthetic_images_u8(4, v.image_size, seed=4)
    ref_t = O.clip_text_forward(sd, O.ClipTextConfig(t.vocab, t.ctx, t.width, t.layers, t.heads, t.mlp_dim, t.out_dim), ids)
    ref_v = O.vit_forward(sd, O.VitConfig(v.image_size, v.patch_size, v.width, v.layers, v.heads, v.mlp_dim, v.out_dim), O.preprocess_u8_exact_size(u8))
    try:
        grouped_t, grouped_v = tt.encode_ids(ids), vt.encode_u8(u8.cuda())
        assert _cos_err(grouped_t, ref_t) < 1e-4 and _cos_err(grouped_v, ref_v) < 1e-4
        L.check(lib.mq_tune(b"small_m_grouped", 0))
        tiled_t, tiled_v = tt.encode_ids(ids), vt.encode_u8(u8.cuda())
        assert _cos_err(grouped_t, tiled_t) < 1e-4 and _cos_err(grouped_v, tiled_v) < 1e-4
        assert not torch.equal(grouped_v, tiled_v)          # (the knob does switch families)
    finally:
        L.check(lib.mq_tune(b"small_m_grouped", 320))
