"""Micro-benchmark of the sub-LayerNorm-fold GEMM forms against the forms they extend, at the EVA02-B/16 x 128 shapes (HIP events, interleaved).
python tools/probes/lnrs_bench.py [M]"""
import os
import sys
import torch
from marqo_amd import _lib as L

lib = L.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25216
s = lambda: torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(0)
GLU = 256


def timeit(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def case(N, K, glu):
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g)
    cs = W.float().sum(1).contiguous()
    st = torch.empty(M, 2, device="cuda")
    L.check(lib.mq_row_stats(a.data_ptr(), st.data_ptr(), M, K, 1e-6, s())) if K <= 2048 else None
    ns = (N + 63) // 64
    part = torch.empty(ns, M, 2, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    if glu:
        base = lambda: L.check(lib.mq_gemm_bf16_ln(a.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), cs.data_ptr(), st.data_ptr(), out.data_ptr(), N, M, N, K, L.MQ_EPI_BIAS | GLU, s()))
        new = lambda: L.check(lib.mq_gemm_bf16_lnrs(a.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), cs.data_ptr(), st.data_ptr(), 0, out.data_ptr(), N, M, N, K, L.MQ_EPI_BIAS | GLU, part.data_ptr(), s()))
    else:
        fl = L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL
        base = lambda: L.check(lib.mq_gemm_bf16_rs(a.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), out.data_ptr(), out.data_ptr(), N, M, N, K, fl, part.data_ptr(), s()))
        new = lambda: L.check(lib.mq_gemm_bf16_lnrs(a.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), cs.data_ptr(), st.data_ptr(), out.data_ptr(), out.data_ptr(), N, M, N, K, fl, part.data_ptr(), s()))
    res = {}
    for rep in range(2):
        res.setdefault("base", []).append(timeit(base))
        for e in os.environ.get("EXPS", "0").split(","):
            os.environ["MQ_EXP"] = e
            res.setdefault("lnrs exp=" + e, []).append(timeit(new))
    print(f"M={M} N={N} K={K} {'glu' if glu else 'residual'}:", {k: [round(x, 1) for x in v] for k, v in res.items()})


case(4096, 768, True)
case(768, 2048, False)
case(768, 768, False)


def rope_case(T, W, heads):
    """QKV GEMM: mq_gemm_bf16_ln vs mq_gemm_bf16_ln_rope per tile height (ROPE=1; needs docs/experiments/r06r_rope_in_qkv_epilogue.patch applied)"""
    nseq = M // T
    m, K, N, hs = nseq * T, W, 3 * W, W // heads
    x = torch.randn(m, K, device="cuda", generator=g).to(torch.bfloat16)
    wf = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bf = torch.randn(N, device="cuda", generator=g)
    cs = wf.float().sum(1).contiguous()
    st = torch.empty(m, 2, device="cuda")
    L.check(lib.mq_row_stats(x.data_ptr(), st.data_ptr(), m, K, 1e-6, s()))
    pairs = torch.rand(T, hs // 2, 2, device="cuda", generator=g).to(torch.float16)
    out = torch.empty(m, N, device="cuda", dtype=torch.bfloat16)
    base = lambda: L.check(lib.mq_gemm_bf16_ln(x.data_ptr(), K, wf.data_ptr(), K, bf.data_ptr(), cs.data_ptr(), st.data_ptr(), out.data_ptr(), N, m, N, K, L.MQ_EPI_BIAS, s()))
    new = lambda: L.check(lib.mq_gemm_bf16_ln_rope(x.data_ptr(), K, wf.data_ptr(), K, bf.data_ptr(), cs.data_ptr(), st.data_ptr(), out.data_ptr(), N, m, N, K, pairs.data_ptr(), T, int(os.environ.get('ROPE_COLS', 2 * W)), hs, s()))
    res = {}
    for rep in range(2):
        for mt in (0, 4):
            L.check(lib.mq_tune(b"gemm_mt", mt))
            res.setdefault(f"ln mt={mt}", []).append(round(timeit(base), 1))
            for e in os.environ.get('EXPS', '0').split(','):
                os.environ['MQ_EXP'] = e
                res.setdefault(f"rope mt={mt} exp={e}", []).append(round(timeit(new), 1))
    L.check(lib.mq_tune(b"gemm_mt", 0))
    print(f"QKV m={m} T={T} W={W}:", res)


if os.environ.get("ROPE"):
    rope_case(197, 768, 12)
    rope_case(257, 1024, 16)
