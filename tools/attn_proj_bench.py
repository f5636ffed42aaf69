"""mq_attention_proj (one launch) against the three launches it replaces (mq_attention, the out-projection with the bf16 residual epilogue and its
row sums, mq_row_stats_finalize) at the ViT-B/32 block's shapes; interleaved, medians of HIP-event times.
python tools/attn_proj_bench.py [--nseq 256,128,512] [--rounds 9] [--iters 20]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marqo_amd import _lib as L

W, HEADS, T = 768, 12, 50


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nseq", default="256,128,512,64")
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    for nseq in [int(v) for v in args.nseq.split(",")]:
        rows = nseq * T
        g = torch.Generator(device="cuda").manual_seed(nseq)
        qkv = (torch.randn(rows, 3 * W, device="cuda", generator=g)).to(torch.bfloat16)
        wo = (torch.randn(W, W, device="cuda", generator=g) / W ** 0.5).to(torch.bfloat16)
        bias = 0.1 * torch.randn(W, device="cuda", generator=g)
        x = torch.randn(rows, W, device="cuda", generator=g).to(torch.bfloat16)
        a = torch.empty(rows, W, device="cuda", dtype=torch.bfloat16)
        part = torch.empty(12, rows, 2, device="cuda")
        stats = torch.empty(rows, 2, device="cuda")
        # something that evicts the operands from the L2s between launches the way the block's other GEMMs do
        spoil_a = torch.randn(rows, 3072, device="cuda").to(torch.bfloat16)
        spoil_w = torch.randn(768, 3072, device="cuda").to(torch.bfloat16)

        def three():
            L.check(lib.mq_attention(qkv.data_ptr(), a.data_ptr(), None, nseq, T, T, W, HEADS, 0, s))
            L.check(lib.mq_gemm_bf16_rs(a.data_ptr(), W, wo.data_ptr(), W, bias.data_ptr(), x.data_ptr(), x.data_ptr(), W, rows, W, W, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL,
                                        part.data_ptr(), s))
            L.check(lib.mq_row_stats_finalize(part.data_ptr(), 12, stats.data_ptr(), rows, W, 1e-5, s))

        def one():
            L.check(lib.mq_attention_proj(qkv.data_ptr(), wo.data_ptr(), bias.data_ptr(), x.data_ptr(), stats.data_ptr(), nseq, T, W, HEADS, 1e-5, None, 0, None, 0, s))

        def spoil():
            L.check(lib.mq_gemm_bf16(spoil_a.data_ptr(), 3072, spoil_w.data_ptr(), 3072, None, None, a.data_ptr(), W, rows, W, 3072, 0, s))

        def timed(fn, with_spoil):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tot = 0.0
            for _ in range(args.iters):
                if with_spoil:
                    spoil()
                e0.record()
                fn()
                e1.record()
                e1.synchronize()
                tot += e0.elapsed_time(e1)
            return tot / args.iters * 1e3

        for fn in (three, one):
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        for with_spoil in (False, True):
            res = {"three": [], "one": []}
            for _ in range(args.rounds):
                res["three"].append(timed(three, with_spoil))
                res["one"].append(timed(one, with_spoil))
            m3, m1 = statistics.median(res["three"]), statistics.median(res["one"])
            print(f"nseq={nseq:4d} T={T} {'behind another GEMM' if with_spoil else 'back to back      '}: three launches {m3:7.1f} us   one launch {m1:7.1f} us   ({(m1 / m3 - 1) * 100:+.1f} %)",
                  flush=True)


if __name__ == "__main__":
    main()
