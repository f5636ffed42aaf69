"""Follow-up to tools/gemm_insitu_probe.py (which found that only a COLD OUTPUT slows a tower GEMM down): how large may the set of
distinct output buffers in rotation be before the penalty appears?  footprint = A + W + n x out.   python tools/gemm_outset_probe.py"""
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
    for name, M, N, K, flags in [("b32 qkv", 12800, 2304, 768, L.MQ_EPI_BIAS), ("b32 fc1", 12800, 3072, 768, L.MQ_EPI_BIAS | L.MQ_EPI_GELU)]:
        W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda", generator=g)
        A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        big = torch.zeros(12, M, N, device="cuda", dtype=torch.bfloat16)          # ONE allocation: the tower's workspace is one, too
        reader = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)

        def timed(n_out, read_back, reps=96, warm=24):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for i in range(warm + reps):
                out = big[i % n_out]
                if i >= warm:
                    ev[i - warm][0].record()
                L.check(lib.mq_gemm_bf16(A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), 0, out.data_ptr(), N, M, N, K, flags, s))
                if i >= warm:
                    ev[i - warm][1].record()
                if read_back:
                    reader.copy_(out)             # a consumer reads the output (like attention / fc2 do), so the lines are clean-or-shared, recently used
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
            return ts[len(ts) // 2]
        timed(1, False, reps=200)
        line = []
        for n_out in (1, 2, 3, 4, 6, 12):
            for rb in (False, True):
                t = sorted(timed(n_out, rb) for _ in range(3))[1]
                line.append(f"n_out={n_out}{' +reader' if rb else ''}: {t:5.1f} us")
        mb = M * N * 2 / 1e6
        print(f"{name} (out {mb:.0f} MB each, A {M * K * 2 / 1e6:.0f} MB, W {N * K * 2 / 1e6:.1f} MB): " + " | ".join(line), flush=True)


if __name__ == "__main__":
    main()
