#!/bin/bash
# What clock does the chip hold under matrix load?  (1) register-resident MFMA burn (tools/probes/mfma_peak.hip), (2) the tower GEMMs
# in a loop with rocm-smi sampled beside them.
mkdir -p gpurun_out
{
  ./tools/probes/mfma_peak 1 300
  ./tools/probes/mfma_peak 2 300
} > gpurun_out/mfma_peak.txt 2>&1
cat gpurun_out/mfma_peak.txt
python - > gpurun_out/gemm_clock.txt 2>&1 <<'PY'
import subprocess, threading, time, re, sys, os
sys.path.insert(0, os.getcwd())
import torch
from marqo_amd import _lib as L
lib = L.load()
s = torch.cuda.current_stream().cuda_stream
samples = []
stop = False
def sampler():
    while not stop:
        t = time.time()
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        sclk = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
        pw = re.findall(r"Power \(W\): ([\d.]+)", out)
        samples.append((t, sclk[:1], pw[:1]))
th = threading.Thread(target=sampler); th.start()
def loop(name, M, N, K, flags, secs=2.5):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda"); f32 = bool(flags & L.MQ_EPI_OUT_F32)
    o = torch.zeros(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    t0 = time.time(); n = 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    while time.time() - t0 < secs:
        for _ in range(50):
            L.check(lib.mq_gemm_bf16(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), o.data_ptr() if (flags & L.MQ_EPI_RESIDUAL) else None, o.data_ptr(), N, M, N, K, flags, s), "gemm")
        n += 50; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    mine = [x for x in samples if x[0] >= t0 + 0.5]
    print(f"{name}: {us:.1f} us/launch {2*M*N*K/us/1e6:.0f} TF/s over {time.time()-t0:.1f} s; sclk samples {[x[1] for x in mine]} power {[x[2] for x in mine]}", flush=True)
time.sleep(1.0)
print("idle samples", samples[-2:], flush=True)
loop("b32 qkv", 12800, 2304, 768, L.MQ_EPI_BIAS)
loop("b32 fc2", 12800, 768, 3072, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32)
loop("4096^3", 4096, 4096, 4096, 0)
stop = True; th.join()
PY
cat gpurun_out/gemm_clock.txt
