#!/usr/bin/env python
"""Calibration only: what does the vendor library (hipBLASLt / rocBLAS behind torch.nn.functional.linear) reach on the towers' GEMM shapes?
Not a product path — marqo_amd never calls it; the numbers sit next to tools/gemm_bench.py's in profiles/ as the practical ceiling of a
tuned library kernel on the same shapes (plain bias epilogue only: the library has no GELU / residual / LayerNorm-stat epilogues)."""
import torch
import torch.nn.functional as F

SHAPES = [("b32 qkv", 12800, 2304, 768), ("b32 out", 12800, 768, 768), ("b32 fc1", 12800, 3072, 768), ("b32 fc2", 12800, 768, 3072),
          ("l14 qkv", 16448, 3072, 1024), ("l14 out", 16448, 1024, 1024), ("l14 fc1", 16448, 4096, 1024), ("l14 fc2", 16448, 1024, 4096),
          ("4096^3", 4096, 4096, 4096), ("8192^3", 8192, 8192, 8192)]


def main():
    tot = 0.0
    for name, M, N, K in SHAPES:
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
        b = torch.randn(N, device="cuda").to(torch.bfloat16)
        for bias in (None, b):
            for _ in range(5):
                F.linear(A, W, bias)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                F.linear(A, W, bias)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 30
            print(f"{name:8s} M={M:6d} N={N:5d} K={K:5d} {'bias' if bias is not None else 'plain':5s} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s")
            if bias is not None and name.startswith("b32"):
                tot += us
    print(f"b32 layer GEMMs (bias form): {tot:.1f} us/layer")


if __name__ == "__main__":
    main()
