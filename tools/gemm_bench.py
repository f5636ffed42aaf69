"""Micro-benchmark of mq_gemm_bf16 at the shapes the towers launch (and a 4096^3 reference point).
python tools/gemm_bench.py [--iters 50]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marqo_amd import _lib as L

SHAPES = [
    # name, M, N, K, flags
    ("b32 qkv", 12800, 2304, 768, L.MQ_EPI_BIAS),
    ("b32 out", 12800, 768, 768, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32),
    ("b32 fc1", 12800, 3072, 768, L.MQ_EPI_BIAS | L.MQ_EPI_GELU),
    ("b32 fc2", 12800, 768, 3072, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32),
    ("b32 out16", 12800, 768, 768, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL),      # bf16 residual stream form (mq_tune residual_bf16)
    ("b32 fc2_16", 12800, 768, 3072, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL),
    ("b32 patch", 12544, 768, 3072, L.MQ_EPI_OUT_F32),
    ("l14 qkv", 16448, 3072, 1024, L.MQ_EPI_BIAS),
    ("l14 out", 16448, 1024, 1024, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32),
    ("l14 fc1", 16448, 4096, 1024, L.MQ_EPI_BIAS | L.MQ_EPI_GELU),
    ("l14 fc2", 16448, 1024, 4096, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32),
    ("l14x128 qkv", 32896, 3072, 1024, L.MQ_EPI_BIAS),                       # ViT-L/14 at 128 images (bench configs[2]): the bf16 residual stream forms
    ("l14x128 out16", 32896, 1024, 1024, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL),
    ("l14x128 fc1", 32896, 4096, 1024, L.MQ_EPI_BIAS | L.MQ_EPI_GELU),
    ("l14x128 fc2_16", 32896, 1024, 4096, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL),
    ("l14x240 qkv", 61680, 3072, 1024, L.MQ_EPI_BIAS),                       # 240 crops (configs[4])
    ("l14x240 fc2_16", 61680, 1024, 4096, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL),
    ("text qkv", 78848, 1536, 512, L.MQ_EPI_BIAS),                           # CLIP text B/32, 1024 x 77 rows
    ("bert fc1", 78848, 3072, 768, L.MQ_EPI_BIAS | L.MQ_EPI_GELU),           # e5-base, 1024 x 77 rows
    ("epi fc1 f32out", 12800, 3072, 768, L.MQ_EPI_OUT_F32),
    ("epi fc1 bf16", 12800, 3072, 768, 0),
    ("epi fc1 bias", 12800, 3072, 768, L.MQ_EPI_BIAS),
    ("epi fc1 gelu", 12800, 3072, 768, L.MQ_EPI_BIAS | L.MQ_EPI_GELU),
    ("epi fc1 res", 12800, 3072, 768, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32),
    ("4096^3", 4096, 4096, 4096, 0),
    ("8192^3", 8192, 8192, 8192, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--only", default="")
    ap.add_argument("--fp8", action="store_true", help="time mq_gemm_fp8 (K taken as-is; fp8 operands)")
    ap.add_argument("--ab", default="", help="interleaved within-process A/B (cdna_hip_programming.md §5.4 rule 24): "
                    "'name:key=v,key=v;name2:key=v' — every variant is timed in every round, medians are reported")
    ap.add_argument("--rounds", type=int, default=7)
    args = ap.parse_args()
    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    tot_t = tot_f = 0.0
    ab_tot = {}
    for name, M, N, K, flags in SHAPES:
        if args.only and not __import__("re").search(args.only, name):
            continue
        if args.fp8:
            if K % 128:
                continue
            A8 = torch.randint(0, 0x78, (M, K), dtype=torch.uint8, device="cuda")
            W8 = torch.randint(0, 0x78, (N, K), dtype=torch.uint8, device="cuda")
            sa = torch.rand(M, device="cuda") + 0.5
            sw = torch.rand(N, device="cuda") * 1e-4
            b8 = torch.randn(N, device="cuda")
            f32 = bool(flags & L.MQ_EPI_OUT_F32)
            gelu = bool(flags & L.MQ_EPI_GELU)
            fl8 = flags | (L.MQ_EPI_OUT_FP8 if gelu else 0) if flags else L.MQ_EPI_OUT_F32
            o8 = torch.zeros(M, N, device="cuda", dtype=torch.float32 if (fl8 & L.MQ_EPI_OUT_F32) else (torch.uint8 if gelu else torch.bfloat16))
            osc = torch.ones(1, device="cuda")
            res8 = o8 if fl8 & L.MQ_EPI_RESIDUAL else None

            def run8():
                L.check(lib.mq_gemm_fp8(A8.data_ptr(), K, W8.data_ptr(), K, sa.data_ptr(), 1, sw.data_ptr(), b8.data_ptr(), L.ptr(res8),
                                        o8.data_ptr(), N, osc.data_ptr(), 0, M, N, K, fl8, s))
            if args.ab:   # interleaved A/B of mq_tune settings, as for the bf16 GEMM below; the first variant's output is the reference for bit-identity
                variants = []
                for spec in args.ab.split(";"):
                    vname, _, kv = spec.partition(":")
                    variants.append((vname, [(k, int(v)) for k, v in (p.split("=") for p in kv.split(",") if p)]))
                times = {vn: [] for vn, _ in variants}
                outs = {}
                for rnd in range(args.rounds + 1):
                    for vn, kvs in variants:
                        for k, v in kvs:
                            L.check(lib.mq_tune(k.encode(), v))
                        if res8 is None:
                            o8.zero_()
                        run8(); run8()
                        if rnd == 0 and res8 is None:
                            outs[vn] = o8.clone()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(args.iters):
                            run8()
                        e1.record()
                        torch.cuda.synchronize()
                        if rnd:
                            times[vn].append(e0.elapsed_time(e1) * 1e3 / args.iters)
                fl = 2.0 * M * N * K
                med = {vn: sorted(t)[len(t) // 2] for vn, t in times.items()}
                base = med[variants[0][0]]
                same = "" if not outs else "  bits: " + " ".join(f"{vn}={'same' if torch.equal(outs[vn], outs[variants[0][0]]) else 'DIFFER'}" for vn, _ in variants[1:])
                print(f"fp8 {name:10s} M={M:6d} N={N:5d} K={K:5d} " + "  ".join(
                    f"{vn}={med[vn]:7.1f}us ({fl / med[vn] / 1e6:6.0f}TF {100 * (med[vn] / base - 1):+5.1f}%)" for vn, _ in variants) + same)
                continue
            for _ in range(5):
                run8()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run8()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            fl = 2.0 * M * N * K
            print(f"fp8 {name:10s} M={M:6d} N={N:5d} K={K:5d}  {us:9.1f} us  {fl / us / 1e6:8.1f} TF/s")
            if name.startswith("b32") and "patch" not in name:
                tot_t += us; tot_f += fl
            continue
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        f32 = bool(flags & L.MQ_EPI_OUT_F32)
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
        res = out if flags & L.MQ_EPI_RESIDUAL else None

        def run():
            L.check(lib.mq_gemm_bf16(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), L.ptr(res), out.data_ptr(), N, M, N, K, flags, s))
        if args.ab:
            variants = []
            for spec in args.ab.split(";"):
                vname, _, kv = spec.partition(":")
                variants.append((vname, [(k, int(v)) for k, v in (p.split("=") for p in kv.split(",") if p)]))
            times = {vn: [] for vn, _ in variants}
            for rnd in range(args.rounds + 1):
                for vn, kvs in variants:
                    for k, v in kvs:
                        L.check(lib.mq_tune(k.encode(), v))
                    run(); run()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.iters):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    if rnd:  # round 0 is warm-up
                        times[vn].append(e0.elapsed_time(e1) * 1e3 / args.iters)
            fl = 2.0 * M * N * K
            med = {vn: sorted(t)[len(t) // 2] for vn, t in times.items()}
            base = med[variants[0][0]]
            print(f"{name:10s} M={M:6d} N={N:5d} K={K:5d} " + "  ".join(
                f"{vn}={med[vn]:7.1f}us ({fl / med[vn] / 1e6:6.0f}TF {100 * (med[vn] / base - 1):+5.1f}%)" for vn, _ in variants))
            if name.startswith("b32") and "patch" not in name:
                for vn in med:
                    ab_tot[vn] = ab_tot.get(vn, 0.0) + med[vn]
            continue
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        fl = 2.0 * M * N * K
        print(f"{name:10s} M={M:6d} N={N:5d} K={K:5d}  {us:9.1f} us  {fl / us / 1e6:8.1f} TF/s")
        if name.startswith("b32") and "patch" not in name:
            tot_t += us; tot_f += fl
    if ab_tot:
        print("b32 layer GEMMs: " + "  ".join(f"{vn}={t:.1f}us" for vn, t in ab_tot.items()))
    if tot_t:
        print(f"b32 layer GEMMs: {tot_t:.1f} us/layer  {tot_f / tot_t / 1e6:.1f} TF/s")


if __name__ == "__main__":
    main()
