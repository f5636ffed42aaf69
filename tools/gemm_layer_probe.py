"""Third cache-state probe (after gemm_insitu_probe.py / gemm_outset_probe.py: cold weights, a freshly written A and a cold output inside ONE
allocation cost nothing): the ViT-B/32 block's own launch sequence rebuilt step by step, every GEMM timed by its own HIP-event pair —
  alone      each GEMM back to back with itself
  gemms      QKV -> out-proj -> fc1 -> fc2 in the block's order and buffers (12 weight sets in rotation)
  +ln        ... with the two LayerNorms in between
  +attn      ... and the attention kernel: the whole block as the tower launches it
so that the step at which the in-tower penalty (QKV 48 -> 55.6 us, fc1 67 -> 74.8 us) appears names its cause.   python tools/gemm_layer_probe.py"""
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
    n_img, T, W, F, H, NL = 256, 50, 768, 3072, 12, 12
    M = n_img * T
    B, G, R = L.MQ_EPI_BIAS, L.MQ_EPI_GELU, L.MQ_EPI_RESIDUAL

    def w(n, k):
        return (torch.randn(n, k, device="cuda", generator=g) / k ** 0.5).to(torch.bfloat16)
    layers = [dict(qkv=w(3 * W, W), out=w(W, W), fc1=w(F, W), fc2=w(W, F)) for _ in range(NL)]
    bq, bo, b1, b2 = (torch.randn(n, device="cuda", generator=g) * 0.1 for n in (3 * W, W, F, W))
    gam, bet = torch.ones(W, device="cuda"), torch.zeros(W, device="cuda")
    ws = torch.zeros(M * (W + W + F + W), device="cuda", dtype=torch.bfloat16)     # one workspace: x | h | qf | a, like the tower's
    x, h, qf, a = ws[:M * W], ws[M * W:2 * M * W], ws[2 * M * W:2 * M * W + M * F], ws[2 * M * W + M * F:]
    x.copy_((torch.randn(M * W, device="cuda", generator=g) * 0.5).to(torch.bfloat16))

    def gemm(A, lda, Wt, bias, res, out, ldc, N, K, flags):
        L.check(lib.mq_gemm_bf16(A.data_ptr(), lda, Wt.data_ptr(), K, bias.data_ptr(), res.data_ptr() if res is not None else 0, out.data_ptr(), ldc,
                                 M, N, K, flags, s))

    def ln():
        L.check(lib.mq_layernorm_ex(x.data_ptr(), 1, 0, gam.data_ptr(), bet.data_ptr(), h.data_ptr(), 0, M, W, 1e-5, s))

    def attn():
        L.check(lib.mq_attention(qf.data_ptr(), a.data_ptr(), 0, n_img, T, T, W, H, 0, s))
    names = ("qkv", "out", "fc1", "fc2")

    def block(lw, rec, with_ln, with_attn, names=names):
        calls = {"qkv": lambda: gemm(h, W, lw["qkv"], bq, None, qf, 3 * W, 3 * W, W, B), "out": lambda: gemm(a, W, lw["out"], bo, x, x, W, W, W, B | R),
                 "fc1": lambda: gemm(h, W, lw["fc1"], b1, None, qf, F, F, W, B | G), "fc2": lambda: gemm(qf, F, lw["fc2"], b2, x, x, W, W, F, B | R)}
        for nm in names:
            if with_ln and nm in ("qkv", "fc1"):
                ln()
            if with_attn and nm == "out":
                attn()
            if rec is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); calls[nm](); e1.record()
                rec[nm].append((e0, e1))
            else:
                calls[nm]()
        x.mul_(0.5)      # (keeps the stream's magnitude bounded over hundreds of blocks; outside the event pairs)

    def run(with_ln, with_attn, steps=8, names=names, same_weights=False):
        for i in range(2 * NL):
            block(layers[0 if same_weights else i % NL], None, with_ln, with_attn, names)
        rec = {nm: [] for nm in names}
        for i in range(steps * NL):
            block(layers[0 if same_weights else i % NL], rec, with_ln, with_attn, names)
        torch.cuda.synchronize()
        return {nm: sorted(a_.elapsed_time(b_) * 1e3 for a_, b_ in v)[len(v) // 2] for nm, v in rec.items()}

    def alone(nm, reps=96):
        lw = layers[0]
        calls = {"qkv": lambda: gemm(h, W, lw["qkv"], bq, None, qf, 3 * W, 3 * W, W, B), "out": lambda: gemm(a, W, lw["out"], bo, x, x, W, W, W, B | R),
                 "fc1": lambda: gemm(h, W, lw["fc1"], b1, None, qf, F, F, W, B | G), "fc2": lambda: gemm(qf, F, lw["fc2"], b2, x, x, W, W, F, B | R)}
        for _ in range(24):
            calls[nm]()
        ev = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); calls[nm](); e1.record(); ev.append((e0, e1))
            if nm in ("out", "fc2"):
                x.mul_(0.5)
        torch.cuda.synchronize()
        return sorted(a_.elapsed_time(b_) * 1e3 for a_, b_ in ev)[reps // 2]
    run(True, True, steps=4)     # clocks settle
    for rnd in range(2):
        print(f"round {rnd}: alone   " + "  ".join(f"{nm} {alone(nm):6.1f}" for nm in names), flush=True)
        for sub in (("qkv", "fc1"), ("qkv", "fc2"), ("qkv", "out"), ("fc1", "fc2"), ("qkv", "fc1", "fc2"), ("qkv", "out", "fc1")):
            for same in (True, False):
                r = run(False, False, names=sub, same_weights=same)
                print(f"round {rnd}: only {'+'.join(sub):12s} {'one weight set' if same else '12 weight sets'}: " + "  ".join(f"{nm} {r[nm]:6.1f}" for nm in sub), flush=True)
        for label, (wl, wa) in (("gemms  ", (False, False)), ("+ln    ", (True, False)), ("+attn  ", (False, True)), ("+ln+attn", (True, True))):
            r = run(wl, wa)
            print(f"round {rnd}: {label} " + "  ".join(f"{nm} {r[nm]:6.1f}" for nm in names) + f"   sum {sum(r.values()):6.1f} us", flush=True)


if __name__ == "__main__":
    main()
