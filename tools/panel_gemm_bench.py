"""mq_panel_gemm_ln (one workgroup per image) against mq_gemm_bf16_ln (tiled) at the ViT-B/32 block's QKV / fc1 shapes; interleaved, medians of HIP-event times.
python tools/panel_gemm_bench.py [--nseq 256,512,200]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marqo_amd import _lib as L

K, T = 768, 50


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nseq", default="256,512,200,128")
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    for nseq in [int(v) for v in args.nseq.split(",")]:
        rows = nseq * T
        for name, N, flags in (("qkv", 2304, L.MQ_EPI_BIAS), ("fc1", 3072, L.MQ_EPI_BIAS | L.MQ_EPI_GELU)):
            g = torch.Generator(device="cuda").manual_seed(nseq + N)
            x = torch.randn(rows, K, device="cuda", generator=g).to(torch.bfloat16)
            w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
            b = torch.randn(N, device="cuda", generator=g)
            cs = w.float().sum(1).contiguous()
            st = torch.stack([x.float().mean(1), 1.0 / torch.sqrt(x.float().var(1, unbiased=False) + 1e-5)], dim=1).contiguous()
            o = torch.empty(rows, N, device="cuda", dtype=torch.bfloat16)

            def tiled():
                L.check(lib.mq_gemm_bf16_ln(x.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), cs.data_ptr(), st.data_ptr(), o.data_ptr(), N, rows, N, K, flags, s))

            def panel():
                L.check(lib.mq_panel_gemm_ln(x.data_ptr(), w.data_ptr(), b.data_ptr(), cs.data_ptr(), st.data_ptr(), o.data_ptr(), N, nseq, T, N, K, flags, s))

            def timed(fn):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                tot = 0.0
                for _ in range(args.iters):
                    e0.record(); fn(); e1.record(); e1.synchronize()
                    tot += e0.elapsed_time(e1)
                return tot / args.iters * 1e3

            for fn in (tiled, panel):
                for _ in range(3):
                    fn()
            torch.cuda.synchronize()
            res = {"tiled": [], "panel": []}
            for _ in range(args.rounds):
                res["tiled"].append(timed(tiled))
                res["panel"].append(timed(panel))
            mt, mp = statistics.median(res["tiled"]), statistics.median(res["panel"])
            gf = 2.0 * rows * N * K / 1e9
            print(f"nseq={nseq:4d} {name} N={N}: tiled {mt:7.1f} us ({gf / mt * 1e3:6.0f} TF/s)   panel {mp:7.1f} us ({gf / mp * 1e3:6.0f} TF/s)   ({(mp / mt - 1) * 100:+.1f} %)", flush=True)


if __name__ == "__main__":
    main()
