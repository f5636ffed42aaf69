"""Where mq_attention_proj spends its time: the launch with phases switched off (MQ_AP_DEBUG bits 1 / 2 / 4 = without the attention phase / the GEMM loop / the
epilogue; results are then meaningless).  NSEQ=256 python tools/attn_proj_phases.py"""
import os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
from marqo_amd import _lib as L
lib = L.load()
W, T = 768, 50
nseq = int(os.environ.get("NSEQ", "256"))
rows = nseq * T
qkv = torch.randn(rows, 3 * W, device="cuda").to(torch.bfloat16)
wo = (torch.randn(W, W, device="cuda") / 27).to(torch.bfloat16)
bias = torch.randn(W, device="cuda")
x = torch.randn(rows, W, device="cuda").to(torch.bfloat16)
st = torch.zeros(rows, 2, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def one():
    L.check(lib.mq_attention_proj(qkv.data_ptr(), wo.data_ptr(), bias.data_ptr(), x.data_ptr(), st.data_ptr(), nseq, T, W, 12, 1e-5, None, 0, None, 0, s))
def timed(iters=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        x.normal_()
        e0.record(); one(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
names = {0: "all", 7: "nothing (launch + setup)", 6: "phase 1 (attention)", 5: "phase 2 (GEMM loop)", 3: "epilogue", 4: "phases 1+2", 1: "2 + epilogue", 2: "1 + epilogue"}
for rep in range(2):
    for m in (0, 7, 6, 5, 3, 4, 1, 2):
        os.environ["MQ_AP_DEBUG"] = str(m)
        one(); torch.cuda.synchronize()
        print(f"nseq={nseq} mode {m} {names[m]:28s} {timed():7.1f} us", flush=True)
