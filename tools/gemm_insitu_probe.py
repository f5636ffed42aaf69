"""Why does a tower GEMM run 11-15 % slower inside the tower than back to back alone?  The same mq_gemm_bf16 launches timed (HIP events on the
launch stream, one event pair per launch) under controlled cache states:
  warm        the same A / W / out every launch (what tools/gemm_bench.py measures: operands sit in the 256 MB Infinity Cache)
  cold_w      12 weight matrices in rotation (a layer's weights were last touched one step ago: HBM), A / out fixed
  fresh_a+..  like the tower: A is WRITTEN by a row-wise kernel (LayerNorm) right before the GEMM; weights in rotation or fixed; out fixed or in rotation
  cold_all    a 1 GB memset between launches (everything from HBM)
  ..prefetched_w  the weights of launch i+1 are read by a small copy kernel enqueued BEFORE launch i (so they sit in the Infinity Cache)
python tools/gemm_insitu_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marqo_amd import _lib as L


def main():
    lib = L.load()
    L.check(lib.mq_tune(b"small_m", 0))
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(7)
    NL = 12
    trash = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    shapes = [("b32 qkv", 12800, 2304, 768, L.MQ_EPI_BIAS, False), ("b32 fc1", 12800, 3072, 768, L.MQ_EPI_BIAS | L.MQ_EPI_GELU, False),
              ("b32 out", 12800, 768, 768, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL, True), ("b32 fc2", 12800, 768, 3072, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL, True)]
    for name, M, N, K, flags, inplace in shapes:
        Ws = [(torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16) for _ in range(NL)]
        bias = torch.randn(N, device="cuda", generator=g)
        x = torch.randn(M, K, device="cuda", generator=g)
        gam, bet = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda")
        As = [torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16) for _ in range(2)]
        outs = [torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(NL)]
        sink = torch.empty(Ws[0].numel(), device="cuda", dtype=torch.bfloat16)

        def gemm(A, W, out):
            L.check(lib.mq_gemm_bf16(A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), out.data_ptr() if inplace else 0, out.data_ptr(), N, M, N, K, flags, s))

        def timed(pre, pick, reps=48, warm=12):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for i in range(warm + reps):
                A, W, out = pick(i)
                pre(i, A)
                if i >= warm:
                    ev[i - warm][0].record()
                gemm(A, W, out)
                if i >= warm:
                    ev[i - warm][1].record()
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
            return ts[len(ts) // 2]

        def ln_into(i, A):
            L.check(lib.mq_layernorm(x.data_ptr(), 0, gam.data_ptr(), bet.data_ptr(), A.data_ptr(), 0, M, K, 1e-5, s))

        def nothing(i, A):
            pass
        variants = {
            "warm": (nothing, lambda i: (As[0], Ws[0], outs[0])),
            "cold_w": (nothing, lambda i: (As[0], Ws[i % NL], outs[0])),
            "fresh_a+cold_w": (ln_into, lambda i: (As[0], Ws[i % NL], outs[0])),          # the tower's situation
            "fresh_a+warm_w": (ln_into, lambda i: (As[0], Ws[0], outs[0])),
            "fresh_a+cold_w+cold_out": (ln_into, lambda i: (As[0], Ws[i % NL], outs[i % NL])),
            "cold_all": (lambda i, A: trash.fill_(i & 255), lambda i: (As[0], Ws[i % NL], outs[i % NL])),
            "fresh_a+prefetched_w": (lambda i, A: (ln_into(i, A), sink.copy_(Ws[(i + 1) % NL].view(-1)))[0], lambda i: (As[0], Ws[i % NL], outs[0])),
        }
        timed(*variants["warm"], reps=200)                       # clocks / power state settle before anything is recorded
        runs = {k: [] for k in variants}
        for rnd in range(3):                                     # three rounds, every variant in each: order effects show up as spread
            for k, (pre, pick) in variants.items():
                runs[k].append(timed(pre, pick, reps=96))
        res = {k: sorted(v)[1] for k, v in runs.items()}
        spread = {k: max(v) - min(v) for k, v in runs.items()}
        fl = 2.0 * M * N * K
        print(f"{name:8s} M={M} N={N} K={K}: " + "  ".join(f"{k} {v:6.1f} us +-{spread[k] / 2:.1f} ({fl / v * 1e-6:5.0f} TF)" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
