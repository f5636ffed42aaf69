"""Bit-identity of mq_gemm_bf16 tile variants against the default narrow tile (mq_tune knobs), every epilogue, ragged shapes.
usage: python tools/gemm_check.py "name:key=v,key=v;name2:..." """
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marqo_amd import _lib as L

lib = L.load()
variants = []
for spec in (sys.argv[1] if len(sys.argv) > 1 else "default:gemm_nh=0;eager:gemm_nh=4;big:gemm_nh=3").split(";"):
    name, _, kv = spec.partition(":")
    variants.append((name, [(k, int(v)) for k, v in (p.split("=") for p in kv.split(",") if p)]))
g = torch.Generator(device="cuda").manual_seed(3)
s = torch.cuda.current_stream().cuda_stream
bad = 0
for (M, N, K) in [(12800, 768, 768), (12800, 3072, 768), (4099, 2304, 768), (16448, 1024, 4096), (700, 260, 64), (256, 256, 128), (5000, 388, 192), (8192, 8192, 1024), (32896, 1024, 1024)]:
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g)
    forms = [(0, None), (L.MQ_EPI_OUT_F32, None), (L.MQ_EPI_BIAS, None), (L.MQ_EPI_BIAS | L.MQ_EPI_GELU, None), (L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32, res),
             (L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL, res.to(torch.bfloat16))]

    def run(f, r):
        out = r.clone() if r is not None else torch.empty(M, N, device="cuda", dtype=torch.float32 if f & L.MQ_EPI_OUT_F32 else torch.bfloat16)
        L.check(lib.mq_gemm_bf16(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), out.data_ptr() if r is not None else 0, out.data_ptr(), N, M, N, K, f, s))
        return out
    for k in ("gemm_nh", "gemm_mt"):
        L.check(lib.mq_tune(k.encode(), 1 if k == "gemm_nh" else 0))
    base = [run(f, r) for f, r in forms]
    for name, kvs in variants:
        for k, v in kvs:
            L.check(lib.mq_tune(k.encode(), v))
        for rep in range(3):
            for (f, r), want in zip(forms, base):
                got = run(f, r)
                if not torch.equal(got, want):
                    bad += 1
                    d = (got.float() - want.float()).abs()
                    print(f"MISMATCH {name} {(M, N, K)} flags={f} rep={rep}: max abs diff {float(d.max()):.3e}, {int((d > 0).sum())} elements")
        L.check(lib.mq_tune(b"gemm_nh", 0))
print("gemm_check:", "all variants bit-identical to the narrow tile" if not bad else f"{bad} mismatches")
