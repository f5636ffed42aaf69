"""The bf16 GEMM main loop (csrc/gemm_bf16.hip: software-pipelined k-loop, persistent tile walk, buffer-descriptor LDS-DMA) at every tile height and
tile order must give the same result as a fp32 reference on the bf16-rounded operands, on ragged shapes (K = 64: one k-step per tile; N = 4), with every
fused epilogue; tile height / tile order change scheduling only, so all of them are bit-identical; repeated launches screen for LDS-ring races (a racy
pipeline shows up as run-to-run differences); a matrix taller than one launch can address goes in row chunks with the same bits."""
import pytest
import torch

from marqo_amd import _lib as L

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _tiled_family(tiled_gemm_only):
    """this file exercises the tiled kernel on every shape, the small ones included (the skinny kernels: tests/test_small_m_gpu.py)"""
    yield


DEFAULTS = dict(gemm_mt=0, gemm_cgroup=8, gemm_nh=0, gemm_tail=0)
SHAPES = [(50, 64, 64), (257, 768, 128), (1000, 132, 192), (4097, 2304, 768), (12800, 768, 3072), (333, 3072, 64), (16, 4, 64),
          (20000, 1160, 64), (3000, 1288, 128)]


def _tune(lib, **kw):
    for k, v in kw.items():
        L.check(lib.mq_tune(k.encode(), v), "mq_tune")


def _gemm(lib, A, W, bias, res, flags, lda=None):
    M, K = A.shape
    N = W.shape[0]
    f32 = bool(flags & L.MQ_EPI_OUT_F32)
    out = res.clone() if (flags & L.MQ_EPI_RESIDUAL) else torch.empty(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    L.check(lib.mq_gemm_bf16(A.data_ptr(), lda or A.stride(0), W.data_ptr(), K, L.ptr(bias), out.data_ptr() if flags & L.MQ_EPI_RESIDUAL else 0,
                             out.data_ptr(), N, M, N, K, flags, torch.cuda.current_stream().cuda_stream))
    return out


@pytest.mark.parametrize("cgroup", [0, 8])
@pytest.mark.parametrize("mt", [0, 2, 4, 5, 6])
def test_every_tile_height_matches_reference(mt, cgroup):
    lib = L.load()
    try:
        _tune(lib, gemm_mt=mt, gemm_cgroup=cgroup)
        g = torch.Generator(device="cuda").manual_seed(mt * 7 + cgroup)
        for (M, N, K) in SHAPES:
            A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
            bias = torch.randn(N, device="cuda", generator=g)
            res = torch.randn(M, N, device="cuda", generator=g)
            ref = A.float() @ W.float().t()
            for flags, want in ((0, ref), (L.MQ_EPI_OUT_F32, ref), (L.MQ_EPI_BIAS, ref + bias),
                                (L.MQ_EPI_BIAS | L.MQ_EPI_GELU, torch.nn.functional.gelu(ref + bias)),
                                (L.MQ_EPI_BIAS | L.MQ_EPI_QUICKGELU, (ref + bias) * torch.sigmoid(1.702 * (ref + bias))),
                                (L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32, ref + bias + res)):
                out = _gemm(lib, A, W, bias, res, flags).float()
                tol = 2e-2 if not (flags & L.MQ_EPI_OUT_F32) else 2e-3
                err = (out - want).abs().max().item() / (want.abs().max().item() + 1e-6)
                assert err < tol, (mt, cgroup, (M, N, K), flags, err)
            res16 = res.to(torch.bfloat16)      # the bf16 residual stream: read-modify-write in bf16
            out = _gemm(lib, A, W, bias, res16, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL).float()
            want = ref + bias + res16.float()
            assert (out - want).abs().max().item() / (want.abs().max().item() + 1e-6) < 2e-2
    finally:
        _tune(lib, **DEFAULTS)


def test_tile_height_and_order_are_bitwise_equal_and_stable():
    """same k-order of MFMAs per output element whatever the tile height and the tile walk -> bit-identical; 25 launches under load screen races"""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(5)
    try:
        for (M, N, K) in [(12800, 768, 768), (4099, 384, 3072), (12800, 2304, 768), (700, 260, 64)]:
            A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
            b = torch.randn(N, device="cuda", generator=g)
            _tune(lib, gemm_mt=4, gemm_cgroup=0)
            base = _gemm(lib, A, W, b, None, L.MQ_EPI_BIAS | L.MQ_EPI_OUT_F32)
            for mt in (0, 2, 5, 6):
                for cg in (0, 4, 8):
                    _tune(lib, gemm_mt=mt, gemm_cgroup=cg)
                    for _ in range(25 if (mt, cg) == (0, 8) else 2):
                        out = _gemm(lib, A, W, b, None, L.MQ_EPI_BIAS | L.MQ_EPI_OUT_F32)
                        assert torch.equal(out, base), (mt, cg, (M, N, K))
    finally:
        _tune(lib, **DEFAULTS)


def test_rows_do_not_depend_on_their_call():
    """a row's result does not depend on which other rows share its call (requests are merged and split freely): sub-ranges of a big call, at offsets
    that are no multiple of any tile height, carry the same bits"""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(6)
    M, N, K = 5000, 1536, 512
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g)
    big = _gemm(lib, A, W, b, None, L.MQ_EPI_BIAS | L.MQ_EPI_GELU)
    for lo, hi in ((0, 1), (333, 1033), (4999, 5000), (2500, 5000)):
        part = _gemm(lib, A[lo:hi].contiguous(), W, b, None, L.MQ_EPI_BIAS | L.MQ_EPI_GELU)
        assert torch.equal(part.view(torch.int16), big[lo:hi].view(torch.int16)), (lo, hi)


def test_a_matrix_taller_than_one_launch_can_address_goes_in_row_chunks():
    """the LDS-DMA addresses its operands through 32-bit buffer offsets: rows x leading dimension x 2 B must stay below 4 GiB per launch, a taller A
    runs as several launches over whole row tiles.  Provoked with a huge leading dimension (a strided view: 9 000 rows, 600 KB apart = 5.4 GB of address
    range, 9 000 x 128 elements actually touched)."""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(7)
    M, N, K, lda = 9000, 256, 128, 300_000
    store = torch.zeros(M * lda + K, device="cuda", dtype=torch.bfloat16)       # 5.4 GB of address range
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    store.as_strided((M, K), (lda, 1)).copy_(A)
    W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g)
    want = _gemm(lib, A, W, b, None, L.MQ_EPI_BIAS)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    L.check(lib.mq_gemm_bf16(store.data_ptr(), lda, W.data_ptr(), K, b.data_ptr(), 0, out.data_ptr(), N, M, N, K, L.MQ_EPI_BIAS,
                             torch.cuda.current_stream().cuda_stream))
    assert torch.equal(out.view(torch.int16), want.view(torch.int16))
    res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
    want2 = _gemm(lib, A, W, b, res, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL)
    out2 = res.clone()
    L.check(lib.mq_gemm_bf16(store.data_ptr(), lda, W.data_ptr(), K, b.data_ptr(), out2.data_ptr(), out2.data_ptr(), N, M, N, K,
                             L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL, torch.cuda.current_stream().cuda_stream))
    assert torch.equal(out2.view(torch.int16), want2.view(torch.int16))


def test_row_chunks_under_a_lowered_address_limit_are_bit_identical():
    """the same chunking exercised at a small size (mq_tune gemm_addr_limit_mb): 5 000 x 768 bf16 = 7.7 MB of A under a 2 MB limit = 4 launches"""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(23)
    M, N, K = 5000, 772, 768
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g)
    forms = [(L.MQ_EPI_BIAS | L.MQ_EPI_GELU, None), (L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32, res), (L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL, res.to(torch.bfloat16))]
    whole = [_gemm(lib, A, W, b, r, f) for f, r in forms]
    try:
        _tune(lib, gemm_addr_limit_mb=2)
        chunked = [_gemm(lib, A, W, b, r, f) for f, r in forms]
    finally:
        _tune(lib, gemm_addr_limit_mb=0)
    for x, y in zip(whole, chunked):
        assert torch.equal(x, y)


@pytest.mark.parametrize("nh,tail", [(0, 0), (4, 0), (3, 0), (4, 1), (3, 1)])
def test_the_big_tile_is_bit_identical_to_the_narrow_one_except_its_tail(nh, tail):
    """round 5: the 256 x 256 8-wave tile — on every row (gemm_nh = 3) and in the row-split plans (gemm_nh = 0 default, 4 eager) — accumulates every
    output element over k in the same order as the (32*MT) x 128 tiles: same bits with every epilogue, on ragged shapes (N not a multiple of 256,
    N < 256 falls back to the narrow tile, K = 64: one k-step per tile); row statistics and the folded LayerNorm included; 20 repeated launches screen
    the LDS ring and the tail's flags for races.  The rows behind the last full 256-row tile go through the in-kernel TAIL (K cut over the grid,
    fp32 partials added in range order): deterministic, and equal to the narrow tile's rows up to the fp32 association of the k-sum — one bf16 ulp
    after the store.  The tail is opt-in (gemm_tail = 1): by default a ragged last row tile is a tile like any other and EVERY row is bit-identical."""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(31)
    try:
        for (M, N, K) in [(12800, 768, 768), (4099, 2304, 768), (16448, 1024, 4096), (700, 260, 64), (224, 256, 128), (5000, 132, 192), (9000, 3072, 1024),
                          (16448, 4096, 1024), (12800, 3072, 768), (32896, 1024, 4096)]:
            A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
            b = torch.randn(N, device="cuda", generator=g)
            res = torch.randn(M, N, device="cuda", generator=g)
            forms = [(0, None), (L.MQ_EPI_OUT_F32, None), (L.MQ_EPI_BIAS | L.MQ_EPI_GELU, None), (L.MQ_EPI_BIAS | L.MQ_EPI_QUICKGELU, None),
                     (L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32, res), (L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL, res.to(torch.bfloat16))]
            _tune(lib, gemm_nh=1)
            base = [_gemm(lib, A, W, b, r, f) for f, r in forms]
            _tune(lib, gemm_nh=nh, gemm_tail=tail)
            full = (M // 256) * 256 if tail else M          # rows in front of the tail
            first = None
            for rep in range(20 if (M, N, K) in ((12800, 768, 768), (16448, 1024, 4096)) else 2):
                big = [_gemm(lib, A, W, b, r, f) for f, r in forms]
                for x, y, (f, _) in zip(base, big, forms):
                    assert torch.equal(x[:full], y[:full]), (nh, tail, (M, N, K), f, rep)
                    if full < M:
                        d = (x[full:].float() - y[full:].float()).abs().max().item()
                        tol = (2 ** -7 if y.dtype == torch.bfloat16 else 2e-5) * (x[full:].float().abs().max().item() + 1e-6)
                        assert d <= tol, (nh, (M, N, K), f, rep, d, tol)
                if first is None:
                    first = big
                else:                        # the tail is deterministic: run to run the same bits
                    assert all(torch.equal(p, q) for p, q in zip(first, big)), (nh, (M, N, K), rep)
    finally:
        _tune(lib, **DEFAULTS)


@pytest.mark.parametrize("M,F,K,ldc_mult", [(4099, 192, 128, 2), (12800, 2048, 768, 2), (300, 64, 64, 1), (1000, 2752, 1024, 2)])
def test_gated_epilogue_forms_the_swiglu_product(M, F, K, ldc_mult):
    """round 6, MQ_EPI_GLU: W = (up, gate) rows interleaved 16 by 16 -> out[m, u] = (A W_up^T + b_up) * silu(A W_gate^T + b_gate), written at row stride ldc
    (2 F inside the towers: the product takes the place of the (up | gate) tensor's first half); every tile height, the big tile, ragged M"""
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(M + F)
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    Wu = (torch.randn(F, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    Wg = (torch.randn(F, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bu, bg = torch.randn(F, device="cuda", generator=g), torch.randn(F, device="cuda", generator=g)
    il = lambda u, v: torch.stack([u.reshape(F // 16, 16, *u.shape[1:]), v.reshape(F // 16, 16, *v.shape[1:])], dim=1).reshape(2 * F, *u.shape[1:]).contiguous()
    W, b = il(Wu, Wg), il(bu, bg)
    up = A.float() @ Wu.float().t() + bu
    gt = A.float() @ Wg.float().t() + bg
    want = up * torch.nn.functional.silu(gt)
    ldc = ldc_mult * F
    GLU = 256
    outs = []
    try:
        for kw in (dict(), dict(gemm_mt=2), dict(gemm_mt=6), dict(gemm_nh=3)):
            _tune(lib, **{**DEFAULTS, **kw})
            out = torch.full((M, ldc), 7.0, device="cuda", dtype=torch.bfloat16)
            L.check(lib.mq_gemm_bf16(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), 0, out.data_ptr(), ldc, M, 2 * F, K, L.MQ_EPI_BIAS | GLU, torch.cuda.current_stream().cuda_stream))
            err = (out[:, :F].float() - want).abs().max().item() / (want.abs().max().item() + 1e-6)
            assert err < 2e-2, (kw, err)
            if ldc > F:
                assert bool((out[:, F:] == 7.0).all())          # nothing is written behind the product
            outs.append(out)
        assert all(torch.equal(o, outs[0]) for o in outs[1:])   # tile shapes change scheduling only
    finally:
        _tune(lib, **DEFAULTS)
