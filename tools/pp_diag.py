"""Diagnostic for csrc/gemm_pp.hip: per epilogue, compare with the shipped kernel and describe WHERE the results differ
(NaN counts, mismatching rows / columns folded into the 128 x 256 tile and the 16 x 16 sub-tile grid).  python tools/pp_diag.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marqo_amd import _lib as L


def main():
    lib = L.load()
    L.check(lib.mq_tune(b"small_m", 0))
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(302)
    for (M, N, K) in [(12800, 2304, 768), (1000, 512, 512)]:
        A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda", generator=g)
        for flags in (L.MQ_EPI_BIAS, L.MQ_EPI_BIAS | L.MQ_EPI_GELU, L.MQ_EPI_BIAS | L.MQ_EPI_QUICKGELU):
            def run():
                out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
                L.check(lib.mq_gemm_bf16(A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), 0, out.data_ptr(), N, M, N, K, flags, s))
                return out
            L.check(lib.mq_tune(b"gemm_pp", 0))
            base = run()
            for pps in (2, 4):
                L.check(lib.mq_tune(b"gemm_pp", 2)); L.check(lib.mq_tune(b"gemm_pp_pps", pps))
                for it in range(3):
                    out = run()
                    bad = (out.view(torch.int16) != base.view(torch.int16))
                    nb = int(bad.sum())
                    line = f"M={M} N={N} K={K} flags={flags} pps={pps} it={it}: base NaN {int(base.float().isnan().sum())} out NaN {int(out.float().isnan().sum())} mismatching {nb}"
                    if nb:
                        idx = bad.nonzero()
                        r, c = idx[:, 0], idx[:, 1]
                        line += (f" | rows {int(r.min())}..{int(r.max())} cols {int(c.min())}..{int(c.max())}"
                                 f" | row%128 hist(16-row groups) {torch.bincount((r % 128) // 16, minlength=8).tolist()}"
                                 f" | col%256 hist(32-col pairs) {torch.bincount((c % 256) // 32, minlength=8).tolist()}"
                                 f" | tiles (row/128) {torch.unique(r // 128)[:12].tolist()} (col/256) {torch.unique(c // 256).tolist()}"
                                 f" | first {idx[:4].tolist()} out {out[r[0], c[0]].item()} base {base[r[0], c[0]].item()}")
                    print(line, flush=True)
    L.check(lib.mq_tune(b"gemm_pp", 0))


if __name__ == "__main__":
    main()
