#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k attention > gpurun_out/attn_waves_tests.log 2>&1; tail -3 gpurun_out/attn_waves_tests.log
python - > gpurun_out/attn_waves_ab.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from marqo_amd import _lib as L
lib = L.load(); s = torch.cuda.current_stream().cuda_stream
def bench(name, nseq, T, heads, hs, causal=False, iters=30, rounds=5):
    W = heads * hs
    qkv = torch.randn(nseq * T, 3 * W, device="cuda").to(torch.bfloat16)
    out = torch.empty(nseq * T, W, device="cuda", dtype=torch.bfloat16)
    res = {}
    for rnd in range(rounds + 1):
        for nw in (4, 8):
            L.check(lib.mq_tune(b"attn_waves", nw))
            run = lambda: L.check(lib.mq_attention(qkv.data_ptr(), out.data_ptr(), None, nseq, T, T, W, heads, L.MQ_MASK_CAUSAL if causal else L.MQ_MASK_NONE, s))
            run(); run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): run()
            e1.record(); torch.cuda.synchronize()
            if rnd: res.setdefault(nw, []).append(e0.elapsed_time(e1) * 1e3 / iters)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    fl = 4.0 * nseq * heads * T * T * hs
    print(f"{name:28s} 4 waves {med[4]:8.1f} us ({fl/med[4]/1e6:6.0f} TF)   8 waves {med[8]:8.1f} us ({fl/med[8]/1e6:6.0f} TF)  {100*(med[8]/med[4]-1):+6.1f}%", flush=True)
bench("ViT-H/14 64x257 h16 d96", 64, 257, 16, 96)
bench("ViT-bigG/14 32x257 h16 d112", 32, 257, 16, 112)
bench("ViT-L/14 64x257 h16 d64", 64, 257, 16, 64)
bench("ViT-L/14-336 32x577 h16 d64", 32, 577, 16, 64)
bench("ViT-B/32 256x50 h12 d64", 256, 50, 12, 64)
bench("text 1024x77 h8 d64 causal", 1024, 77, 8, 64, True)
bench("BERT 256x512 h12 d64", 256, 512, 12, 64)
L.check(lib.mq_tune(b"attn_waves", 0))
PY
cat gpurun_out/attn_waves_ab.txt
