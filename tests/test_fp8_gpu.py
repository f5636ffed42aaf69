"""K13: fp8 (OCP e4m3) path.  The fp8 GEMM must reproduce, to fp32-accumulation accuracy, the exact product of the
DEQUANTISED operands (this pins the MFMA operand mapping, swizzle, scales and epilogues independent of quantisation
error); quantisation kernels must round like torch.float8_e4m3fn; the fp8 tower is then compared with the fp32 oracle and
its cosine error REPORTED against the bf16 path (north-star tolerance 1e-3 is stated for bf16; fp8 is config 5)."""
import ctypes as C

import pytest
import torch

from marqo_amd import _lib as L

pytestmark = pytest.mark.gpu


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _quant_rows(lib, Wb):
    N, K = Wb.shape
    W8 = torch.empty(N, K, dtype=torch.uint8, device="cuda")
    sc = torch.empty(N, dtype=torch.float32, device="cuda")
    L.check(lib.mq_quantize_weights_fp8(Wb.data_ptr(), K, W8.data_ptr(), K, sc.data_ptr(), N, K, _stream()))
    return W8, sc


def _deq(q8, scale):
    return q8.view(torch.float8_e4m3fn).float() * (scale[:, None] if scale.ndim == 1 and scale.numel() == q8.shape[0] else scale)


def test_quantize_weights_matches_torch_e4m3():
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(0)
    Wb = (torch.randn(300, 768, device="cuda", generator=g) * torch.rand(300, 1, device="cuda", generator=g) * 3).to(torch.bfloat16)
    Wb[7] = 0
    W8, sc = _quant_rows(lib, Wb)
    ref_sc = Wb.float().abs().amax(1) / 448.0
    ref_sc[7] = 1.0
    assert torch.allclose(sc, ref_sc, rtol=1e-6)
    ref8 = (Wb.float() / sc[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    assert (W8 != ref8).float().mean().item() < 1e-3  # ties may differ by one code when x/s vs x*(1/s) differ in the last bit
    err = (_deq(W8, sc) - Wb.float()).abs().amax(1) / (Wb.float().abs().amax(1) + 1e-9)
    assert err.max().item() <= 2 ** -4 + 1e-3


def test_layernorm_fp8_rowscale():
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(1)
    for W in (512, 768, 1024):
        x = torch.randn(1000, W, device="cuda", generator=g) * 3 + 0.5
        gam = torch.rand(W, device="cuda", generator=g) + 0.5
        bet = torch.randn(W, device="cuda", generator=g) * 0.1
        q = torch.empty(1000, W, dtype=torch.uint8, device="cuda")
        s = torch.empty(1000, device="cuda")
        f = torch.empty(1000, W, device="cuda")
        L.check(lib.mq_layernorm_fp8(x.data_ptr(), gam.data_ptr(), bet.data_ptr(), q.data_ptr(), s.data_ptr(), f.data_ptr(), 1000, W, 1e-5, _stream()))
        ref = torch.nn.functional.layer_norm(x, (W,), gam, bet, 1e-5)
        assert torch.allclose(f, ref, atol=2e-5)
        assert torch.allclose(s, ref.abs().amax(1) / 448, rtol=1e-4)
        deq = _deq(q, s)
        assert ((deq - ref).abs().amax(1) / ref.abs().amax(1)).max().item() <= 2 ** -4 + 1e-3


def test_layernorm_fp8_from_the_bf16_stream():
    """the fp8 LayerNorm of an fp8 tower whose residual stream is bf16: the same codes and scales as the fp32-input kernel on the same values"""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(11)
    for W in (512, 768, 1024, 1280):
        x16 = (torch.randn(777, W, device="cuda", generator=g) * 3 + 0.5).to(torch.bfloat16)
        gam = torch.rand(W, device="cuda", generator=g) + 0.5
        bet = torch.randn(W, device="cuda", generator=g) * 0.1
        q, s = torch.empty(777, W, dtype=torch.uint8, device="cuda"), torch.empty(777, device="cuda")
        q2, s2 = torch.empty_like(q), torch.empty_like(s)
        L.check(lib.mq_layernorm_fp8_ex(x16.data_ptr(), 1, gam.data_ptr(), bet.data_ptr(), q.data_ptr(), s.data_ptr(), 0, 777, W, 1e-5, _stream()))
        x32 = x16.float()
        L.check(lib.mq_layernorm_fp8(x32.data_ptr(), gam.data_ptr(), bet.data_ptr(), q2.data_ptr(), s2.data_ptr(), 0, 777, W, 1e-5, _stream()))
        assert torch.equal(q, q2) and torch.equal(s, s2)
        f = torch.empty(4, W, device="cuda")
        assert lib.mq_layernorm_fp8_ex(x16.data_ptr(), 1, gam.data_ptr(), bet.data_ptr(), q.data_ptr(), s.data_ptr(), f.data_ptr(), 4, W, 1e-5, _stream()) != 0


@pytest.mark.parametrize("mt", [0, 2, 4, 5, 6])
def test_gemm_fp8_equals_product_of_dequantised_operands(mt):
    lib = L.load()
    L.check(lib.mq_tune(b"gemm_mt", mt))
    try:
        g = torch.Generator(device="cuda").manual_seed(2 + mt)
        for (M, N, K) in [(50, 64, 128), (257, 768, 256), (1000, 132, 384), (4097, 2304, 768), (12800, 768, 3072), (16, 4, 128),
                          (300, 176, 128), (300, 1160, 256), (20000, 1168, 128)]:
            A8 = torch.randint(0, 256, (M, K), dtype=torch.uint8, device="cuda", generator=g)
            A8[(A8 & 0x7F) == 0x7F] = 0x30  # no NaN codes (0x7F / 0xFF)
            W8 = torch.randint(0, 256, (N, K), dtype=torch.uint8, device="cuda", generator=g)
            W8[(W8 & 0x7F) == 0x7F] = 0x30
            # keep magnitudes moderate: clear the top exponent bit
            A8 &= 0xBF; W8 &= 0xBF
            sa = torch.rand(M, device="cuda", generator=g) + 0.5
            sw = torch.rand(N, device="cuda", generator=g) + 0.5
            bias = torch.randn(N, device="cuda", generator=g)
            res = torch.randn(M, N, device="cuda", generator=g)
            ref = (_deq(A8, sa).double() @ _deq(W8, sw).double().t())
            scal = torch.tensor([0.75], device="cuda")
            ref_s = ((A8.view(torch.float8_e4m3fn).double() * 0.75) @ _deq(W8, sw).double().t())
            tol = 3e-4 * ref.abs().max().item() + 1e-6  # the MX MFMA aligns the 128 products of a block before adding them

            def run(flags, rowscale, out_dtype, residual=None, out_scale=None, amax=None):
                out = residual.clone() if residual is not None else torch.empty(M, N, device="cuda", dtype=out_dtype)
                L.check(lib.mq_gemm_fp8(A8.data_ptr(), K, W8.data_ptr(), K, (sa if rowscale else scal).data_ptr(), 1 if rowscale else 0,
                                        sw.data_ptr(), bias.data_ptr(), out.data_ptr() if residual is not None else 0, out.data_ptr(), N,
                                        L.ptr(out_scale), L.ptr(amax), M, N, K, flags, _stream()))
                return out

            o = run(L.MQ_EPI_OUT_F32, True, torch.float32)
            assert (o.double() - ref).abs().max().item() < tol, (mt, M, N, K)
            o = run(L.MQ_EPI_OUT_F32, False, torch.float32)
            assert (o.double() - ref_s).abs().max().item() < 3e-4 * ref_s.abs().max().item() + 1e-6
            o = run(L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32, True, torch.float32, residual=res)
            assert (o.double() - (ref + bias + res)).abs().max().item() < tol + 1e-4
            o = run(L.MQ_EPI_BIAS, True, torch.bfloat16)
            assert (o.double() - (ref + bias)).abs().max().item() < 1e-2 * (ref.abs().max().item() + 3)
            # bf16 residual stream: read-modify-write of bf16 rows (fp32 sum, one bf16 rounding at the store)
            res16 = res.to(torch.bfloat16)
            o = run(L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL, True, torch.bfloat16, residual=res16)
            want16 = ref + bias + res16.double()
            assert (o.double() - want16).abs().max().item() < 2 ** -8 * want16.abs().max().item() + tol + 1e-4, (mt, M, N, K)
            want = torch.nn.functional.gelu((ref + bias).float())
            osc = (want.abs().max() / 448).reshape(1)
            amax = torch.zeros(1, device="cuda")
            o = run(L.MQ_EPI_BIAS | L.MQ_EPI_GELU | L.MQ_EPI_OUT_FP8, True, torch.uint8, out_scale=osc, amax=amax)
            assert abs(amax.item() - want.abs().max().item()) < 1e-3 * want.abs().max().item() + 1e-4
            deq = o.view(torch.float8_e4m3fn).float() * osc
            assert (deq - want).abs().max().item() <= (2 ** -4) * want.abs().max().item() + 1e-3
    finally:
        L.check(lib.mq_tune(b"gemm_mt", 0))


def test_gemm_fp8_scheduling_knobs_are_bit_identical():
    """the L2-blocked tile order changes scheduling only: every epilogue form
    must give bit-identical results with the knobs off and on, on shapes with ragged right edges and many tiles per workgroup"""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(99)
    knobs_off = dict(gemm_cgroup=0)
    knobs_on = dict(gemm_cgroup=8)
    try:
        for (M, N, K) in [(12800, 3072, 768), (12800, 2304, 768), (20000, 1168, 128), (300, 176, 256), (5000, 1552, 384)]:
            A8 = torch.randint(0, 256, (M, K), dtype=torch.uint8, device="cuda", generator=g) & 0xBF
            W8 = torch.randint(0, 256, (N, K), dtype=torch.uint8, device="cuda", generator=g) & 0xBF
            sa = torch.rand(M, device="cuda", generator=g) + 0.5
            sw = (torch.rand(N, device="cuda", generator=g) + 0.5) * 0.01
            bias = torch.randn(N, device="cuda", generator=g)
            osc = torch.tensor([0.05], device="cuda")
            outs = {}
            for name, knobs in (("off", knobs_off), ("on", knobs_on)):
                for k, v in knobs.items():
                    L.check(lib.mq_tune(k.encode(), v))
                res = []
                for flags, dt in ((L.MQ_EPI_BIAS, torch.bfloat16), (L.MQ_EPI_BIAS | L.MQ_EPI_GELU | L.MQ_EPI_OUT_FP8, torch.uint8),
                                  (L.MQ_EPI_OUT_F32, torch.float32)):
                    out = torch.zeros(M, N, device="cuda", dtype=dt)
                    amax = torch.zeros(1, device="cuda")
                    L.check(lib.mq_gemm_fp8(A8.data_ptr(), K, W8.data_ptr(), K, sa.data_ptr(), 1, sw.data_ptr(), bias.data_ptr(), 0, out.data_ptr(), N,
                                            osc.data_ptr(), amax.data_ptr(), M, N, K, flags, _stream()))
                    res += [out, amax.clone()]
                outs[name] = res
            for a, b in zip(outs["off"], outs["on"]):
                assert torch.equal(a, b), (M, N, K, a.dtype)
    finally:
        for k, v in knobs_on.items():
            L.check(lib.mq_tune(k.encode(), v))


def test_gemm_fp8_taller_than_one_launch_can_address_goes_in_row_chunks():
    """the fp8 GEMM addresses its operands through 32-bit buffer offsets like the bf16 one; an A above the limit must run as row chunks
    (per-row scales, residual and output offset per chunk), not be refused.  The limit is lowered (mq_tune gemm_addr_limit_mb) so that 3 MB of A
    need 4 launches; every epilogue form must be bit-identical to the single launch."""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(17)
    M, N, K = 3000, 388, 1024
    A8 = torch.randint(0, 256, (M, K), dtype=torch.uint8, device="cuda", generator=g) & 0xBF
    W8 = torch.randint(0, 256, (N, K), dtype=torch.uint8, device="cuda", generator=g) & 0xBF
    sa = torch.rand(M, device="cuda", generator=g) + 0.5
    sw = (torch.rand(N, device="cuda", generator=g) + 0.5) * 0.01
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g)
    osc = torch.tensor([0.05], device="cuda")

    def run_all():
        outs = []
        for flags, dt, residual in ((L.MQ_EPI_BIAS, torch.bfloat16, None), (L.MQ_EPI_BIAS | L.MQ_EPI_GELU | L.MQ_EPI_OUT_FP8, torch.uint8, None),
                                    (L.MQ_EPI_OUT_F32, torch.float32, None), (L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32, torch.float32, res),
                                    (L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL, torch.bfloat16, res.to(torch.bfloat16))):
            for rowscale in (1, 0):
                out = residual.clone() if residual is not None else torch.zeros(M, N, device="cuda", dtype=dt)
                amax = torch.zeros(1, device="cuda")
                L.check(lib.mq_gemm_fp8(A8.data_ptr(), K, W8.data_ptr(), K, sa.data_ptr(), rowscale, sw.data_ptr(), bias.data_ptr(),
                                        out.data_ptr() if residual is not None else 0, out.data_ptr(), N, osc.data_ptr(), amax.data_ptr(), M, N, K, flags, _stream()))
                outs += [out, amax.clone()]
        return outs
    whole = run_all()
    try:
        L.check(lib.mq_tune(b"gemm_addr_limit_mb", 1))
        chunked = run_all()
        # a weight above the limit cannot be chunked: refused, loudly
        big_w = torch.zeros(1200, K, dtype=torch.uint8, device="cuda")
        out = torch.zeros(M, 1200, device="cuda")
        rc = lib.mq_gemm_fp8(A8.data_ptr(), K, big_w.data_ptr(), K, sa.data_ptr(), 1, torch.ones(1200, device="cuda").data_ptr(), 0, 0, out.data_ptr(), 1200, 0, 0,
                             M, 1200, K, L.MQ_EPI_OUT_F32, _stream())
        assert rc != 0
    finally:
        L.check(lib.mq_tune(b"gemm_addr_limit_mb", 0))
    for a, b in zip(whole, chunked):
        assert torch.equal(a, b), a.dtype


def _cos_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((1 - (a * b).sum(-1) / (a.norm(dim=-1) * b.norm(dim=-1))).max())


def test_fp8_towers_vs_oracle_and_bf16():
    """ViT-B/32-shaped towers at reduced depth (6 layers keeps the CPU oracle quick): the fp8 path must stay close to
    the fp32 oracle; the bound asserted here (1e-2) is the honest fp8 figure, the measured values are printed."""
    from marqo_amd.engine import archs, synthetic, towers
    from oracle import towers as O
    varch = archs.VitArch(224, 32, 768, 6, 12, 3072, 512)
    tarch = archs.ClipTextArch(49408, 77, 512, 6, 8, 2048, 512)
    sd = synthetic.random_open_clip_state_dict(vision=varch, text=tarch, seed=0)
    u8 = O.synthetic_images_u8(24, 224, seed=3).cuda()
    ref = O.vit_forward(sd, O.VitConfig(224, 32, 768, 6, 12, 3072, 512), O.preprocess_u8_exact_size(u8.cpu()))
    bf = towers.VitTower(varch, sd, "cuda:0").encode_u8(u8)
    t8 = towers.VitTower(varch, sd, "cuda:0", precision="fp8")
    t8.calibrate_fp8(lambda: t8.encode_u8(u8[:16]))
    f8 = t8.encode_u8(u8)
    e_bf, e_f8 = _cos_err(bf, ref), _cos_err(f8, ref)
    print(f"ViT 6L: 1-cos vs fp32 oracle  bf16 {e_bf:.2e}  fp8 {e_f8:.2e}")
    assert e_bf < 3e-4 and e_f8 < 1e-2
    assert torch.equal(t8.encode_u8(u8), f8)  # frozen scales -> deterministic
    ids = O.synthetic_clip_ids(16, seed=4)
    reft = O.clip_text_forward(sd, O.ClipTextConfig(49408, 77, 512, 6, 8, 2048, 512), ids)
    tt8 = towers.ClipTextTower(tarch, sd, "cuda:0", precision="fp8")
    tt8.calibrate_fp8(lambda: tt8.encode_ids(ids))
    e_t8 = _cos_err(tt8.encode_ids(ids), reft)
    print(f"CLIP text 6L: 1-cos vs fp32 oracle  fp8 {e_t8:.2e}")
    assert e_t8 < 1e-2
    with pytest.raises(RuntimeError):
        towers.VitTower(varch, sd, "cuda:0").calibrate_fp8(lambda: None)


def test_rowquant_fp8_matches_torch():
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(21)
    x = torch.randn(333, 768, device="cuda", generator=g) * torch.rand(333, 1, device="cuda", generator=g) * 5
    x[5] = 0.0
    q = torch.empty(333, 768, dtype=torch.uint8, device="cuda")
    sc = torch.empty(333, device="cuda")
    L.check(lib.mq_rowquant_fp8(x.data_ptr(), q.data_ptr(), sc.data_ptr(), 333, 768, _stream()))
    want_sc = x.abs().amax(1) / 448
    want_sc[5] = 1.0
    assert torch.allclose(sc, want_sc, rtol=1e-6)
    ref = (x / want_sc[:, None]).to(torch.float8_e4m3fn)
    assert torch.equal(q.view(torch.float8_e4m3fn).float(), ref.float())


def test_fp8_bert_vs_oracle_and_bf16():
    """post-LN (BERT) encoder on the fp8 path: every LayerNorm rewrites the fp32 stream and leaves the e4m3 operand of the next
    GEMM.  BERT-base shape at 4 layers, ragged packed sequences, mean and CLS pooling."""
    from marqo_amd.engine import archs, towers
    from oracle import towers as O
    cfg = O.BertConfig(vocab=30522, max_pos=512, width=768, layers=4, heads=12, mlp_dim=3072)
    arch = archs.BertArch(vocab=30522, max_pos=512, width=768, layers=4, heads=12, mlp_dim=3072)
    sd = O.synthetic_bert_state_dict(cfg, seed=1)
    g = torch.Generator().manual_seed(3)
    n, S = 24, 40
    lens = torch.randint(3, S + 1, (n,), generator=g)
    ids = torch.randint(1000, 30522, (n, S), generator=g)
    mask = (torch.arange(S)[None, :] < lens[:, None]).long()
    ids = ids * mask
    for pooling in ("mean", "cls"):
        cfg.pooling = pooling
        ref = O.hf_encode(sd, cfg, ids, mask)
        bf = towers.BertTower(arch, sd, "cuda:0", pooling=pooling).encode_ids(ids, mask)
        t8 = towers.BertTower(arch, sd, "cuda:0", pooling=pooling, precision="fp8")
        t8.calibrate_fp8(lambda: t8.encode_ids(ids[:16], mask[:16]))
        f8 = t8.encode_ids(ids, mask)
        e_bf, e_f8 = _cos_err(bf, ref), _cos_err(f8, ref)
        print(f"BERT 4L {pooling}: 1-cos vs fp32 oracle  bf16 {e_bf:.2e}  fp8 {e_f8:.2e}")
        assert e_bf < 3e-4 and e_f8 < 1e-2
        assert torch.equal(t8.encode_ids(ids, mask), f8)


# ---- full depth: the fp8 policy (bf16 first blocks, e4m3 last blocks) calibrated at load ----------------------------------------------
def _vit_l14():
    from marqo_amd.engine import archs
    from oracle import towers as O
    varch, _ = archs.resolve_open_clip("ViT-L-14")
    return varch, O.VitConfig(224, 14, 1024, 24, 16, 4096, 768)


@pytest.mark.parametrize("weights", ["plain", "realistic"])
def test_fp8_policy_full_depth_vit_l14_meets_1e3(weights):
    """BASELINE configs[4]'s tower at its real depth (24 blocks), on benign random weights and on the trained-like fixture (LayerNorm
    gain spread, outlier channels, peaky attention, a class-token massive activation): the load-time policy must land inside the
    1e-3 north-star tolerance against the fp32 CPU oracle, deterministically, with a non-trivial share of the blocks on fp8; running
    EVERY block on fp8 is measured too and printed (the honest all-fp8 bound, > 1e-3 by design of the format)."""
    from marqo_amd.engine import towers
    from oracle import towers as O
    varch, ocfg = _vit_l14()
    sd = O.synthetic_vit_state_dict(ocfg, 0) if weights == "plain" else O.synthetic_vit_state_dict_realistic(ocfg, 0)
    u8 = O.synthetic_images_u8(2, 224, seed=5)
    ref = O.vit_forward(sd, ocfg, O.preprocess_u8_exact_size(u8))
    bf = towers.VitTower(varch, sd, "cuda:0").encode_u8(u8.cuda())
    t8 = towers.VitTower(varch, sd, "cuda:0", precision="fp8")
    first = t8.tune_fp8_default()
    f8 = t8.encode_u8(u8.cuda())
    e_bf, e_f8 = _cos_err(bf.cpu(), ref), _cos_err(f8.cpu(), ref)
    print(f"ViT-L/14 24L {weights}: 1-cos vs fp32 oracle  bf16 {e_bf:.2e}  fp8 policy (blocks {first}..23 on e4m3) {e_f8:.2e}  "
          f"[calibration batch vs bf16: policy {t8.fp8_calibration_error:.2e}, all 24 blocks {t8.fp8_all_blocks_error:.2e}]")
    assert e_bf < 3e-4
    assert e_f8 < 1e-3
    assert first <= 20, "the policy should keep at least the last few blocks on fp8"
    assert torch.equal(t8.encode_u8(u8.cuda()), f8)                       # frozen scales + frozen split: deterministic
    t8b = towers.VitTower(varch, sd, "cuda:0", precision="fp8")
    assert t8b.tune_fp8_default() == first and torch.equal(t8b._fp8.scale, t8._fp8.scale)   # same policy on every load
    assert t8b.fp8_mlp_extra == t8.fp8_mlp_extra and t8b.cfg.enc.fp8_mlp_extra == t8.fp8_mlp_extra
    assert torch.equal(t8b.encode_u8(u8.cuda()), f8)
    # all blocks on fp8 (budget = infinity): still a valid, deterministic mode; its error is what the format costs
    assert t8b.tune_fp8_default(budget=1.0) == 0
    e_all = _cos_err(t8b.encode_u8(u8.cuda()).cpu(), ref)
    print(f"ViT-L/14 24L {weights}: every block on e4m3: 1-cos vs fp32 oracle {e_all:.2e}")
    assert e_all < 2e-2


def test_fp8_policy_chunked_images_cfg5():
    """BASELINE configs[4]: ViT-L/14 fp8 + on-GPU image chunking (3x3 'simple' grid, 10 crops per image) — every crop embedding within
    1e-3 of the fp32 oracle run on the oracle's own (Pillow-exact) crops"""
    import numpy as np
    from marqo_amd.engine import towers
    from marqo_amd.engine.preprocess import ImagePreprocessor
    from oracle import preprocess as OP
    from oracle import towers as O
    varch, ocfg = _vit_l14()
    sd = O.synthetic_vit_state_dict_realistic(ocfg, 1)
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (300, 420, 3), dtype=np.uint8)
    u8, boxes = ImagePreprocessor("cuda:0", 224).chunk_grid_u8([img], 3, 3, False)
    t8 = towers.VitTower(varch, sd, "cuda:0", precision="fp8")
    t8.tune_fp8_default()
    emb = t8.encode_u8(u8).cpu()
    patches, bbs = OP.chunk_image_simple(img, 3, 3, False)
    ref = O.vit_forward(sd, ocfg, torch.from_numpy(np.stack([OP.clip_transform(p, 224) for p in patches])))
    assert emb.shape == (10, 768) and np.allclose(boxes[0], np.asarray(bbs), rtol=1e-6)
    e = _cos_err(emb, ref)
    print(f"cfg 5 (ViT-L/14 fp8 policy, 10 crops): 1-cos vs fp32 oracle {e:.2e}, blocks {t8.fp8_first_layer}..23 on e4m3")
    assert e < 1e-3


def test_default_policies_together_stay_inside_the_tolerance_on_held_out_natural_crops():
    """the load-time policies stack — bf16 residual stream (<= 5e-4 vs the fp32 stream), e4m3 block split (<= 5e-4 vs the bf16
    tower), LayerNorm fold, on-GPU chunker — and each is decided on a seeded U{0..255} calibration batch.  This runs ALL defaults together on the
    trained-like 24-block ViT-L/14 fixture (LN-gamma spread, massive-activation channels) with HELD-OUT inputs of natural-image statistics
    (1/f spectrum, correlated channels, flat regions: oracle.synthetic_natural_images_u8), 26 source images -> 260 crops through the 3 x 3 grid
    chunker, and holds every crop embedding against the fp32 oracle run on the oracle's own Pillow-exact crops: 1 - cos < 1e-3, the north-star
    tolerance.  The chosen policy and its load-time errors are printed.
    Anchor: /root/reference/tests/core/inference/embedding_models/test_hugging_face_model.py:614-634 (embeddings within tolerance of stored vectors)."""
    import numpy as np
    from marqo_amd.engine import towers
    from marqo_amd.engine.preprocess import ImagePreprocessor
    from oracle import preprocess as OP
    from oracle import towers as O
    varch, ocfg = _vit_l14()
    sd = O.synthetic_vit_state_dict_realistic(ocfg, 2)
    src = O.synthetic_natural_images_u8(26, 360, 480, seed=5).numpy()
    pre = ImagePreprocessor("cuda:0", 224)
    t8 = towers.VitTower(varch, sd, "cuda:0", precision="fp8")
    first = t8.tune_fp8_default()
    crops, _ = pre.chunk_grid_u8([src[i] for i in range(len(src))], 3, 3, False)
    emb = t8.encode_u8(crops).cpu()
    t16 = towers.VitTower(varch, sd, "cuda:0")          # the bf16 tower with ITS defaults (stream policy + LayerNorm fold) on the same crops
    emb16 = t16.encode_u8(crops).cpu()
    ref_crops = []
    for i in range(len(src)):
        patches, _ = OP.chunk_image_simple(src[i], 3, 3, False)
        ref_crops.extend(OP.clip_transform(p, 224) for p in patches)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(min(nthreads, 16))      # (the GPU boxes grant 16 CPUs of a 256-thread host: more threads than that only spin)
    try:
        ref = torch.cat([O.vit_forward(sd, ocfg, torch.from_numpy(np.stack(ref_crops[k:k + 16]))) for k in range(0, len(ref_crops), 16)])
    finally:
        torch.set_num_threads(nthreads)
    assert emb.shape == (260, 768) and ref.shape == (260, 768)
    cos = lambda a, b: 1 - torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=-1)
    e8, e16 = cos(emb, ref), cos(emb16, ref)
    print(f"defaults stacked, ViT-L/14 realistic weights, 260 natural-statistics crops: fp8 policy (blocks {first}..23 on e4m3, +{t8.fp8_mlp_extra} MLP-only, "
          f"stream {t8.residual_stream}; at load: {t8.fp8_calibration_error:.2e} vs bf16) max 1-cos vs fp32 oracle {float(e8.max()):.2e} "
          f"(median {float(e8.median()):.2e}); bf16 tower (stream {t16.residual_stream}, at load {t16.residual_stream_error}) max {float(e16.max()):.2e}")
    assert float(e16.max()) < 5e-4
    assert float(e8.max()) < 1e-3


def test_fp8_policy_post_ln_and_text_towers():
    """the split also exists for the causal CLIP text tower and the post-LN BERT encoder (bf16 blocks hand a bf16 operand to the first
    e4m3 block through mq_rowquant_fp8): full registry depth (12 blocks), policy inside 1e-3"""
    from marqo_amd.engine import archs, towers
    from oracle import towers as O
    _, tarch = archs.resolve_open_clip("ViT-L-14")
    tcfg = O.ClipTextConfig(49408, 77, 768, 12, 12, 3072, 768)
    sd = O.synthetic_clip_text_state_dict_realistic(tcfg, 0)
    ids = O.synthetic_clip_ids(16, seed=4)
    ref = O.clip_text_forward(sd, tcfg, ids)
    tt = towers.ClipTextTower(tarch, sd, "cuda:0", precision="fp8")
    first = tt.tune_fp8_default()
    e = _cos_err(tt.encode_ids(ids).cpu(), ref)
    print(f"CLIP text L/14 12L realistic: fp8 policy blocks {first}..11, 1-cos vs fp32 oracle {e:.2e} (all blocks vs bf16: {tt.fp8_all_blocks_error:.2e})")
    assert e < 1e-3
    bcfg = O.BertConfig(vocab=30522, max_pos=512, width=768, layers=12, heads=12, mlp_dim=3072)
    barch = archs.BertArch(vocab=30522, max_pos=512, width=768, layers=12, heads=12, mlp_dim=3072)
    bsd = O.synthetic_bert_state_dict(bcfg, seed=1)
    bids, mask = O.synthetic_bert_batch(16, 8, 40, seed=2)
    bref = O.hf_encode(bsd, bcfg, bids, mask)
    for forced in (None, 5):   # the calibrated split, and a forced mid-stack split that exercises the bf16 -> e4m3 hand-over
        tb = towers.BertTower(barch, bsd, "cuda:0", precision="fp8")
        first = tb.tune_fp8_default()
        if forced is not None:
            tb.cfg.enc.fp8_first_layer = forced
        e = _cos_err(tb.encode_ids(bids, mask).cpu(), bref)
        print(f"BERT-base 12L: fp8 blocks {tb.cfg.enc.fp8_first_layer}..11, 1-cos vs fp32 oracle {e:.2e}")
        assert e < (1e-3 if forced is None else 5e-3)


def test_fp8_mlp_only_blocks_in_front_of_the_split():
    """mq_encoder_cfg.fp8_mlp_extra (ABI 6): the blocks in front of the split run only their MLP half on e4m3.  A forced (split, extra) is a
    different, deterministic computation from (split, 0) and from (split - extra, 0), its error against the bf16 tower lies between theirs
    (less e4m3 work than the one, more than the other) up to noise, a bad combination is refused, and the 2-D policy search never returns
    a configuration outside its budget nor one with a smaller e4m3 share than the one-dimensional split."""
    from marqo_amd.engine import towers
    from oracle import towers as O
    varch, ocfg = _vit_l14()
    sd = O.synthetic_vit_state_dict_realistic(ocfg, 0)
    t8 = towers.VitTower(varch, sd, "cuda:0", precision="fp8")
    cal = t8.calibration_images()
    t8.calibrate_fp8(lambda: t8.encode_u8(cal), passes=2, margin=t8.FP8_SCALE_MARGIN)
    bf = towers.VitTower(varch, sd, "cuda:0").encode_u8(cal)
    enc = t8.cfg.enc

    def run(first, extra):
        enc.fp8_first_layer, enc.fp8_mlp_extra = first, extra
        return t8.encode_u8(cal)
    a, b, c = run(16, 0), run(16, 12), run(4, 0)
    assert torch.equal(run(16, 12), b) and not torch.equal(a, b) and not torch.equal(b, c)
    ea, eb, ec = _cos_err(a.cpu(), bf.cpu()), _cos_err(b.cpu(), bf.cpu()), _cos_err(c.cpu(), bf.cpu())
    print(f"ViT-L/14 realistic, 1-cos vs the bf16 tower: split 16 {ea:.2e} | split 16 + 12 MLP-only blocks {eb:.2e} | split 4 {ec:.2e}")
    assert ea * 0.7 < eb < ec * 1.3 and eb < 3e-3
    enc.fp8_first_layer, enc.fp8_mlp_extra = 4, 5
    with pytest.raises(Exception, match="fp8_mlp_extra"):
        t8.encode_u8(cal)
    first = t8.tune_fp8_default()
    assert t8.fp8_calibration_error <= t8.FP8_BUDGET and 0 <= t8.fp8_mlp_extra <= first
    one_d = [tr for tr in t8.fp8_policy_trace if tr[1] == 0][0]
    assert (24 - first) + 2 / 3 * t8.fp8_mlp_extra >= (24 - one_d[0]) - 1e-9
    print(f"policy: split {first}, MLP-only blocks {t8.fp8_mlp_extra}, error {t8.fp8_calibration_error:.2e}; trace {t8.fp8_policy_trace}")


def test_fp8_tower_on_the_bf16_residual_stream(tiled_gemm_only, monkeypatch):
    """An fp8 tower may keep its residual stream in bf16 (decided inside tune_fp8, out of the SAME budget: every error is measured against
    the fp32-stream bf16 run): forced on and off here — both inside 1e-3 of the fp32 oracle, different computations, each deterministic;
    the pooled-rows-only last block stays dead-row elimination (bit-identical to all rows) on the bf16 stream too; and the automatic choice
    takes the bf16 stream only when it costs at most FP8_STREAM_SHARE of the budget."""
    from marqo_amd.engine import towers
    from oracle import towers as O
    varch, ocfg = _vit_l14()
    sd = O.synthetic_vit_state_dict(ocfg, 0)
    u8 = O.synthetic_images_u8(3, 224, seed=6)
    ref = O.vit_forward(sd, ocfg, O.preprocess_u8_exact_size(u8))
    outs = {}
    for mode in ("fp32", "bf16", "auto"):
        monkeypatch.setenv("MARQO_AMD_RESIDUAL_STREAM", mode)
        t8 = towers.VitTower(varch, sd, "cuda:0", precision="fp8")
        first = t8.tune_fp8_default()
        out = t8.encode_u8(u8.cuda())
        assert torch.equal(t8.encode_u8(u8.cuda()), out)
        e = _cos_err(out.cpu(), ref)
        print(f"fp8 ViT-L/14, MARQO_AMD_RESIDUAL_STREAM={mode}: stream {t8.residual_stream} (alone {t8.residual_stream_error}), split {first} + "
              f"{t8.fp8_mlp_extra} MLP-only, 1-cos vs fp32 oracle {e:.2e}, calibration vs the fp32-stream bf16 run {t8.fp8_calibration_error:.2e}")
        assert e < 1e-3 and t8.fp8_calibration_error <= t8.FP8_BUDGET
        assert t8.cfg.enc.residual_stream == (1 if t8.residual_stream == "bf16" else 2)
        if mode != "auto":
            assert t8.residual_stream == mode
        else:
            assert t8.residual_stream == "fp32" or t8.residual_stream_error <= t8.FP8_STREAM_SHARE * t8.FP8_BUDGET
        outs[mode] = out
        if mode == "bf16":
            try:
                L.check(L.load().mq_tune(b"row_select", 0))
                full = t8.encode_u8(u8.cuda())
            finally:
                L.check(L.load().mq_tune(b"row_select", 1))
            assert torch.equal(full, out)
    assert not torch.equal(outs["fp32"], outs["bf16"])


@pytest.mark.parametrize("M,N,K", [(4099, 768, 256), (1000, 260, 128), (16448, 1024, 4096)])
def test_big_8_wave_tile_is_bit_identical_to_the_narrow_tiles(M, N, K):
    """round 6: the 192 x 256 x 128 tile of 8 waves (gemm_fp8_kernel NH = 2, WM = 4) — same k-order per output element as the (32 MT) x 128 tiles, so the
    same bits, with every epilogue: per-row scales + bias -> bf16; GELU -> e4m3 (+ amax); bias + bf16 residual in place; bias + fp32 residual; plain fp32;
    ragged M and N (N % 256 != 0, N < 256 falls back to the narrow tile by itself)"""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A8 = torch.randint(0, 0x78, (M, K), dtype=torch.uint8, device="cuda", generator=g)
    W8 = torch.randint(0, 0x78, (N, K), dtype=torch.uint8, device="cuda", generator=g)
    sa_row = torch.rand(M, device="cuda", generator=g) + 0.5
    sa_one = torch.tensor([0.7], device="cuda")
    sw = torch.rand(N, device="cuda", generator=g) * 1e-3
    bias = torch.randn(N, device="cuda", generator=g)
    res32 = torch.randn(M, N, device="cuda", generator=g)
    osc = torch.tensor([0.05], device="cuda")
    forms = [(L.MQ_EPI_BIAS, 1, torch.bfloat16, None), (L.MQ_EPI_BIAS | L.MQ_EPI_GELU | L.MQ_EPI_OUT_FP8, 1, torch.uint8, None),
             (L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL, 0, torch.bfloat16, res32.to(torch.bfloat16)), (L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32, 0, torch.float32, res32),
             (L.MQ_EPI_OUT_F32, 0, torch.float32, None)]

    def run(flags, rowscale, dtype, res):
        out = res.clone() if res is not None else torch.empty(M, N, device="cuda", dtype=dtype)
        amax = torch.zeros(1, device="cuda")
        L.check(lib.mq_gemm_fp8(A8.data_ptr(), K, W8.data_ptr(), K, (sa_row if rowscale else sa_one).data_ptr(), rowscale, sw.data_ptr(), bias.data_ptr(),
                                out.data_ptr() if res is not None else 0, out.data_ptr(), N, osc.data_ptr(), amax.data_ptr(), M, N, K, flags, _stream()))
        return out, amax
    try:
        L.check(lib.mq_tune(b"gemm_nh", 1))
        narrow = [run(*f) for f in forms]
        L.check(lib.mq_tune(b"gemm_nh", 3))
        for f, (want, want_amax) in zip(forms, narrow):
            for _ in range(2):
                got, amax = run(*f)
                assert torch.equal(got.view(torch.uint8) if got.dtype == torch.uint8 else got, want), f[0]
                assert torch.equal(amax, want_amax)
    finally:
        L.check(lib.mq_tune(b"gemm_nh", 0))
