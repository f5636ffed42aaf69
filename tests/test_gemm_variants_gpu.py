"""Every GEMM variant (2-stage, persistent tile loop with cross-tile prefetch, L2-blocked tile order, widened bf16 stores,
CU-sized tile) x tile height must give the same result as a fp32 reference on the bf16-rounded operands, on ragged shapes,
with every fused epilogue; repeated launches screen for LDS-ring races (a racy pipeline shows up as run-to-run differences)."""
import ctypes as C

import pytest
import torch

from marqo_amd import _lib as L

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _tiled_family(tiled_gemm_only):
    """this file exercises the tiled kernels on every shape, the small ones included (the skinny kernels: tests/test_small_m_gpu.py)"""
    yield

VARIANTS = [dict(gemm_persist=0, gemm_cgroup=0, gemm_wide=0), dict(gemm_persist=1, gemm_cgroup=0, gemm_wide=0),
            dict(gemm_persist=0, gemm_cgroup=8, gemm_wide=1), dict(gemm_persist=1, gemm_cgroup=4, gemm_wide=2)]
DEFAULTS = dict(gemm_mt=0, gemm_persist=1, gemm_cgroup=8, gemm_wide=2, gemm_big=0, gemm_k32=0)


def _vid(v):
    return "p%dc%dw%d" % (v["gemm_persist"], v["gemm_cgroup"], v["gemm_wide"])
SHAPES = [(50, 64, 64), (257, 768, 128), (1000, 132, 192), (4097, 2304, 768), (12800, 768, 3072), (333, 3072, 64), (16, 4, 64),
          (20000, 1160, 64), (3000, 1288, 128)]


def _tune(lib, **kw):
    for k, v in kw.items():
        L.check(lib.mq_tune(k.encode(), v), "mq_tune")


def _gemm(lib, A, W, bias, res, flags):
    M, K = A.shape
    N = W.shape[0]
    f32 = bool(flags & L.MQ_EPI_OUT_F32)
    out = res.clone() if (flags & L.MQ_EPI_RESIDUAL) else torch.empty(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    L.check(lib.mq_gemm_bf16(A.data_ptr(), K, W.data_ptr(), K, L.ptr(bias), out.data_ptr() if flags & L.MQ_EPI_RESIDUAL else 0,
                             out.data_ptr(), N, M, N, K, flags, torch.cuda.current_stream().cuda_stream))
    return out


@pytest.mark.parametrize("variant", VARIANTS, ids=_vid)
@pytest.mark.parametrize("mt", [0, 2, 4, 5, 6])
def test_variant_matches_reference(variant, mt):
    lib = L.load()
    try:
        _tune(lib, gemm_mt=mt, **variant)
        g = torch.Generator(device="cuda").manual_seed(mt * 7 + variant["gemm_persist"])
        for (M, N, K) in SHAPES:
            A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
            bias = torch.randn(N, device="cuda", generator=g)
            res = torch.randn(M, N, device="cuda", generator=g)
            ref = A.float() @ W.float().t()
            for flags, want in ((0, ref), (L.MQ_EPI_OUT_F32, ref), (L.MQ_EPI_BIAS, ref + bias),
                                (L.MQ_EPI_BIAS | L.MQ_EPI_GELU, torch.nn.functional.gelu(ref + bias)),
                                (L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32, ref + bias + res)):
                out = _gemm(lib, A, W, bias, res, flags).float()
                tol = 2e-2 if not (flags & L.MQ_EPI_OUT_F32) else 2e-3
                err = (out - want).abs().max().item() / (want.abs().max().item() + 1e-6)
                assert err < tol, (variant, mt, (M, N, K), flags, err)
    finally:
        _tune(lib, **DEFAULTS)


@pytest.mark.parametrize("variant", VARIANTS[1:], ids=_vid)
def test_variant_is_bitwise_stable_and_equal_to_baseline(variant):
    """same k-order of MFMAs in every variant -> bit-identical to the 2-stage kernel; 25 launches under load screen races"""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(5)
    try:
        for (M, N, K) in [(12800, 768, 768), (4099, 384, 3072), (12800, 2304, 768)]:
            A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
            for mt in (0, 4, 6):
                _tune(lib, gemm_mt=mt, **VARIANTS[0])
                base = _gemm(lib, A, W, None, None, L.MQ_EPI_OUT_F32)
                _tune(lib, gemm_mt=mt, **variant)
                for _ in range(25):
                    out = _gemm(lib, A, W, None, None, L.MQ_EPI_OUT_F32)
                    assert torch.equal(out, base), (variant, mt, (M, N, K))
    finally:
        _tune(lib, **DEFAULTS)


def test_short_kstep_three_workgroup_variant():
    """gemm_k32.hip: 128x128x32 tiles under the 64-byte-row LDS swizzle, three workgroups per CU.  Same k-order of MFMAs as the
    shipped kernel -> bit-identical; ragged shapes, every epilogue, repeated launches as a race screen."""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(77)
    try:
        for (M, N, K) in SHAPES + [(12800, 2304, 768), (12800, 768, 3072)]:
            A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
            bias = torch.randn(N, device="cuda", generator=g)
            res = torch.randn(M, N, device="cuda", generator=g)
            for flags in (0, L.MQ_EPI_OUT_F32, L.MQ_EPI_BIAS, L.MQ_EPI_BIAS | L.MQ_EPI_GELU, L.MQ_EPI_BIAS | L.MQ_EPI_QUICKGELU,
                          L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32):
                _tune(lib, gemm_k32=0)
                base = _gemm(lib, A, W, bias, res, flags)
                _tune(lib, gemm_k32=3)
                for _ in range(4):
                    out = _gemm(lib, A, W, bias, res, flags)
                    assert torch.equal(out, base), ((M, N, K), flags, (out.float() - base.float()).abs().max().item())
    finally:
        _tune(lib, **DEFAULTS)


@pytest.mark.parametrize("big", [4, 5, 6, 8])
def test_cu_sized_tile_variant(big):
    """gemm_big.hip: 8-wave (32*MT)x256 tile, one workgroup per CU"""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(big)
    try:
        for (M, N, K) in SHAPES + [(12800, 2304, 768), (600, 260, 128)]:
            A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
            bias = torch.randn(N, device="cuda", generator=g)
            res = torch.randn(M, N, device="cuda", generator=g)
            _tune(lib, gemm_big=0)
            base = _gemm(lib, A, W, bias, res, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32)
            base_g = _gemm(lib, A, W, bias, res, L.MQ_EPI_BIAS | L.MQ_EPI_GELU)
            _tune(lib, gemm_big=big)
            for _ in range(10):
                assert torch.equal(_gemm(lib, A, W, bias, res, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32), base), (big, M, N, K)
            assert torch.equal(_gemm(lib, A, W, bias, res, L.MQ_EPI_BIAS | L.MQ_EPI_GELU), base_g)
    finally:
        _tune(lib, **DEFAULTS)


def test_counted_vmcnt_and_static_priority_knobs_are_bit_identical():
    """mq_tune("gemm_vmcnt", 1): the first k-step after an epilogue waits only for the stage-0 LDS-DMA (the epilogue's stores stay in
    flight); mq_tune("gemm_prio", 1): static priority for the second workgroup of a CU.  Same arithmetic, same order: identical bits,
    on multi-tile persistent shapes (where the counted wait is actually taken), ragged edges included; repeated as a race screen."""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(91)
    try:
        for (M, N, K) in [(12800, 2304, 768), (12800, 3072, 768), (16448, 4096, 1024), (12801, 2308, 768), (9000, 1540, 192)]:
            A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
            bias = torch.randn(N, device="cuda", generator=g)
            res = torch.randn(M, N, device="cuda", generator=g)
            for flags in (0, L.MQ_EPI_OUT_F32, L.MQ_EPI_BIAS, L.MQ_EPI_BIAS | L.MQ_EPI_GELU, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32):
                _tune(lib, gemm_vmcnt=0, gemm_prio=0)
                base = _gemm(lib, A, W, bias, res, flags)
                for vm, pr in ((1, 0), (0, 1), (1, 1)):
                    _tune(lib, gemm_vmcnt=vm, gemm_prio=pr)
                    for _ in range(6):
                        assert torch.equal(_gemm(lib, A, W, bias, res, flags), base), ((M, N, K), flags, vm, pr)
    finally:
        _tune(lib, gemm_vmcnt=0, gemm_prio=0)
        _tune(lib, **DEFAULTS)


@pytest.mark.parametrize("waves", [8, 4])
@pytest.mark.parametrize("pps", [2, 4])
def test_two_accumulator_set_kernel_is_bit_identical(pps, waves):
    """gemm_pp.hip: one workgroup per CU (8 waves = two per SIMD, or 4 = one per SIMD), 128x256 tiles, 3-stage LDS ring with the barrier in the middle of the k-step, second
    accumulator set (the previous tile's epilogue rides under the next tile's MFMAs, stores / residual loads through buffer descriptors).
    Same k-order of MFMAs and the same epilogue arithmetic as the shipped kernel -> identical bits: every epilogue (fp32 and bf16
    residual streams included), one-tile and many-tile launches, ragged M / N edges, the shortest legal K; repeated as a race screen."""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(300 + pps + waves)
    F = L
    try:
        shapes = [(12800, 2304, 768), (12800, 3072, 768), (12800, 768, 768), (12800, 768, 3072), (16448, 1024, 1024), (12801, 2308, 768),
                  (9000, 1540, 320), (64, 256, 320), (129, 260, 576), (100000, 512, 512), (40000, 4096, 1024)]
        for (M, N, K) in shapes:
            A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
            bias = torch.randn(N, device="cuda", generator=g)
            res = torch.randn(M, N, device="cuda", generator=g)
            res16 = res.to(torch.bfloat16)
            big = M * N > 3e8
            for flags in (0, F.MQ_EPI_OUT_F32, F.MQ_EPI_BIAS | F.MQ_EPI_OUT_F32, F.MQ_EPI_BIAS, F.MQ_EPI_BIAS | F.MQ_EPI_GELU,
                          F.MQ_EPI_BIAS | F.MQ_EPI_QUICKGELU, F.MQ_EPI_BIAS | F.MQ_EPI_RESIDUAL | F.MQ_EPI_OUT_F32,
                          F.MQ_EPI_BIAS | F.MQ_EPI_RESIDUAL):
                if big and flags not in (F.MQ_EPI_BIAS, F.MQ_EPI_BIAS | F.MQ_EPI_RESIDUAL | F.MQ_EPI_OUT_F32):
                    continue

                def run():
                    if flags == (F.MQ_EPI_BIAS | F.MQ_EPI_RESIDUAL):     # bf16 residual stream: in place on a bf16 tensor
                        out = res16.clone()
                        L.check(lib.mq_gemm_bf16(A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), out.data_ptr(), out.data_ptr(), N, M, N, K,
                                                 flags, torch.cuda.current_stream().cuda_stream))
                        return out
                    return _gemm(lib, A, W, bias, res, flags)
                _tune(lib, gemm_pp=0)
                base = run()
                _tune(lib, gemm_pp=2, gemm_pp_pps=pps, gemm_pp_waves=waves)
                for _ in range(2 if big else 5):
                    out = run()
                    assert torch.equal(out, base), ((M, N, K), flags, pps, waves, (out.float() - base.float()).abs().max().item())
    finally:
        _tune(lib, gemm_pp=0, gemm_pp_pps=0, gemm_pp_waves=8, **DEFAULTS)
