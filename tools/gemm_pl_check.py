"""Bit-identity of the software-pipelined GEMM main loop (gemm_pl.hip) against the round 1-3 kernel, over ragged shapes, every epilogue
and every tile height; then an interleaved A/B timing at the towers' shapes.   python tools/gemm_pl_check.py [--no-bench]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marqo_amd import _lib as L


def run(lib, A, W, b, res, out, flags, s):
    M, K = A.shape
    N = W.shape[0]
    L.check(lib.mq_gemm_bf16(A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), b.data_ptr(), L.ptr(res), out.data_ptr(), out.stride(0), M, N, K, flags, s))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-bench", action="store_true")
    args = ap.parse_args()
    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(0)
    B, G, Q, R, F = L.MQ_EPI_BIAS, L.MQ_EPI_GELU, L.MQ_EPI_QUICKGELU, L.MQ_EPI_RESIDUAL, L.MQ_EPI_OUT_F32
    flag_sets = [0, F, B | F, B, B | G, B | Q, B | R | F, B | R]
    shapes = [(12800, 2304, 768), (12800, 768, 768), (1000, 520, 192), (333, 132, 64), (4096, 1024, 1024), (161, 4, 128), (2050, 3072, 128),
              (6400, 768, 3072), (97, 1280, 320)]
    bad = 0
    for (M, N, K) in shapes:
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        for flags in flag_sets:
            f32 = bool(flags & F)
            dt = torch.float32 if f32 else torch.bfloat16
            res0 = torch.randn(M, N, device="cuda").to(dt)
            for mt in (0, 2, 4, 5, 6):
                outs = []
                for pl in (0, 1):
                    L.check(lib.mq_tune(b"gemm_mt", mt))
                    L.check(lib.mq_tune(b"gemm_pl", pl))
                    out = res0.clone() if flags & R else torch.full((M, N), 7.0, device="cuda", dtype=dt)
                    run(lib, A, W, b, out if flags & R else None, out, flags, s)
                    torch.cuda.synchronize()
                    outs.append(out)
                same = torch.equal(outs[0].view(torch.int32 if f32 else torch.int16), outs[1].view(torch.int32 if f32 else torch.int16))
                if not same:
                    bad += 1
                    d = (outs[0].float() - outs[1].float()).abs()
                    print(f"MISMATCH M={M} N={N} K={K} flags={flags:#x} mt={mt}: max|d|={d.max().item():.4g} n={int((d > 0).sum())}")
        # a run of repeats on the big shapes: races show up as run-to-run differences
        if M >= 4096:
            L.check(lib.mq_tune(b"gemm_mt", 0)); L.check(lib.mq_tune(b"gemm_pl", 1))
            ref = None
            for it in range(20):
                out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
                run(lib, A, W, b, None, out, B, s)
                torch.cuda.synchronize()
                if ref is None:
                    ref = out
                elif not torch.equal(ref.view(torch.int16), out.view(torch.int16)):
                    bad += 1
                    print(f"RACE M={M} N={N} K={K}: repeat {it} differs")
                    break
    L.check(lib.mq_tune(b"gemm_mt", 0))
    print("bit-identity:", "OK" if not bad else f"{bad} FAILURES")
    if args.no_bench:
        return 1 if bad else 0
    bench = [("b32 qkv", 12800, 2304, 768, B), ("b32 out16", 12800, 768, 768, B | R), ("b32 fc1", 12800, 3072, 768, B | G), ("b32 fc2_16", 12800, 768, 3072, B | R),
             ("b32 patch", 12544, 768, 3072, F), ("l14 qkv", 16448, 3072, 1024, B), ("l14 out16", 16448, 1024, 1024, B | R), ("l14 fc1", 16448, 4096, 1024, B | G),
             ("l14 fc2_16", 16448, 1024, 4096, B | R), ("text qkv", 78848 // 8, 1536, 512, B), ("4096^3", 4096, 4096, 4096, 0), ("8192^3", 8192, 8192, 8192, 0)]
    variants = [("old", [("gemm_pl", 0)]), ("pl0", [("gemm_pl", 1), ("gemm_pl_ord", 0)]), ("pl1", [("gemm_pl", 1), ("gemm_pl_ord", 1)]), ("pl2", [("gemm_pl", 1), ("gemm_pl_ord", 2)])]
    tot = {vn: 0.0 for vn, _ in variants}
    for name, M, N, K, flags in bench:
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if flags & F else torch.bfloat16)
        res = out if flags & R else None
        times = {vn: [] for vn, _ in variants}
        iters = 10 if M * N * K > 2e11 else 30
        for rnd in range(6):
            for vn, kvs in variants:
                for k, v in kvs:
                    L.check(lib.mq_tune(k.encode(), v))
                run(lib, A, W, b, res, out, flags, s); run(lib, A, W, b, res, out, flags, s)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    run(lib, A, W, b, res, out, flags, s)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    times[vn].append(e0.elapsed_time(e1) * 1e3 / iters)
        fl = 2.0 * M * N * K
        med = {vn: sorted(t)[len(t) // 2] for vn, t in times.items()}
        base = med["old"]
        print(f"{name:10s} M={M:6d} N={N:5d} K={K:5d} " + "  ".join(f"{vn}={med[vn]:7.1f}us ({fl / med[vn] / 1e6:5.0f}TF {100 * (med[vn] / base - 1):+5.1f}%)" for vn, _ in variants))
        if name.startswith("b32") and "patch" not in name:
            for vn in med:
                tot[vn] += med[vn]
    print("b32 layer GEMMs: " + "  ".join(f"{vn}={t:.1f}us" for vn, t in tot.items()))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
