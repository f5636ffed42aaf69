#!/usr/bin/env python
"""Where does a workgroup of gemm_nt_kernel spend its cycles?  Runs the -DMQ_GEMM_TRACE build (tools/probes/build_gemm_trace.sh)
on the towers' GEMM shapes and prints, per shape, the share of each phase of the k-loop in the traced waves' time:
  vmcnt   parked at s_waitcnt vmcnt(0): the next stage's LDS-DMA has not landed yet (load latency not covered)
  barrier parked at the workgroup barrier (skew between the 4 waves)
  body    ds_reads + MFMAs + the next stage's LDS-DMA issues
  epilogue issue time of bias / activation / residual / stores
usage: MARQO_AMD_LIB=tools/probes/libmarqo_hip_trace.so python tools/probes/gemm_trace.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from marqo_amd import _lib as L

SHAPES = [("b32 qkv", 12800, 2304, 768, L.MQ_EPI_BIAS), ("b32 out", 12800, 768, 768, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32),
          ("b32 fc1", 12800, 3072, 768, L.MQ_EPI_BIAS | L.MQ_EPI_GELU), ("b32 fc2", 12800, 768, 3072, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32),
          ("4096^3", 4096, 4096, 4096, 0),
          # epilogue ablation on the QKV shape: no bias / bias / fp32 out / bias+GELU
          ("qkv plain", 12800, 2304, 768, 0), ("qkv bias", 12800, 2304, 768, L.MQ_EPI_BIAS), ("qkv f32out", 12800, 2304, 768, L.MQ_EPI_OUT_F32),
          ("qkv gelu", 12800, 2304, 768, L.MQ_EPI_BIAS | L.MQ_EPI_GELU)]


def main():
    lib = L.load()
    rd = lib.mq_gemm_trace_read
    rd.restype, rd.argtypes = C.c_int, [C.c_void_p]
    s = torch.cuda.current_stream().cuda_stream
    print(f"{'shape':10s} {'us':>7s} {'TF':>6s} | per traced wave: {'k-steps':>7s} {'tiles':>5s} {'cyc/k-step':>10s} | {'vmcnt':>6s} {'barrier':>7s} {'body':>6s} {'epilogue':>8s} (share of the sum)")
    for name, M, N, K, flags in SHAPES:
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        f32 = bool(flags & L.MQ_EPI_OUT_F32)
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
        res = out if flags & L.MQ_EPI_RESIDUAL else None
        run = lambda: L.check(lib.mq_gemm_bf16(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), L.ptr(res), out.data_ptr(), N, M, N, K, flags, s))
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        buf = np.zeros(64 * 4 * 6, dtype=np.uint64)
        assert rd(buf.ctypes.data) == 0
        t = buf.reshape(64, 4, 6).astype(np.float64)
        t = t[t[..., 0] > 0]
        steps, vm, bar, body, epi, tiles = [t[:, i].mean() for i in range(6)]
        tot = vm + bar + body + epi
        if os.environ.get("MQ_TRACE_VERBOSE"):
            print("   per-wave epilogue cycles/tile:", np.round(t[:8, 4] / np.maximum(t[:8, 5], 1)).astype(int).tolist())
        span = np.zeros(2048 * 4, dtype=np.uint64)
        sp = lib.mq_gemm_span_read
        sp.restype, sp.argtypes = C.c_int, [C.c_void_p]
        assert sp(span.ctypes.data) == 0
        span = span.reshape(2048, 4)
        span = span[span[:, 1] > 0]
        t0, t1 = span[:, 0].astype(np.float64) / 100.0, span[:, 1].astype(np.float64) / 100.0      # us (100 MHz counter)
        base = t0.min()
        cu = (span[:, 2] & 0xFFFFFFFF).astype(np.int64)
        cu_key = ((span[:, 2] >> 32).astype(np.int64) & 0xF) * 4096 + ((cu >> 8) & 0xF) + 16 * ((cu >> 12) & 0x1) + 32 * ((cu >> 13) & 0x7)
        per_cu = np.bincount(np.unique(cu_key, return_inverse=True)[1])
        dur = t1 - t0
        print(f"   spans ({len(span)} workgroups on {len(per_cu)} CUs, {per_cu.min()}..{per_cu.max()} per CU): starts 0..{(t0 - base).max():.1f} us "
              f"(p50 {np.median(t0 - base):.1f}), ends {(t1 - base).min():.1f}..{(t1 - base).max():.1f} us (p50 {np.median(t1 - base):.1f}), "
              f"duration mean {dur.mean():.1f} max {dur.max():.1f} us; by tiles: "
              + ", ".join(f"{int(k)} tiles: n={int((span[:, 3] == k).sum())} dur {dur[span[:, 3] == k].mean():.1f} us" for k in np.unique(span[:, 3])))
        print(f"{name:10s} {us:7.1f} {2.0 * M * N * K / us / 1e6:6.0f} | {'':16s} {steps:7.1f} {tiles:5.1f} {(vm + bar + body) / steps:10.0f} | "
              f"{vm / tot:6.1%} {bar / tot:7.1%} {body / tot:6.1%} {epi / tot:8.1%}")


if __name__ == "__main__":
    main()
