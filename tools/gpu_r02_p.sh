#!/bin/bash
# round 2, GPU call P: five-wave attention workgroups for 65..80-token sequences (mq_tune attn_waves -4 = the four-wave form) — tests + text workloads A/B
TAG=${1:-r02p}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_towers_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_sel.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" $OUT/pytest_sel.log | tail -8
for r in 1 2; do
for wl in clip_text_b32 bert_base_77 clip_text_l14; do
  for w in 0 -4; do
    MQ_ATTN_WAVES=$w python - $wl $w <<'PY' >> $OUT/ab.log 2>$OUT/err.txt || tail -3 $OUT/err.txt >> $OUT/ab.log
import sys, os, json, subprocess
wl, w = sys.argv[1], int(sys.argv[2])
sys.argv = ["bench.py", "--workload", wl, "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-extras"]
from marqo_amd import _lib as L
L.check(L.load().mq_tune(b"attn_waves", w))
import io, contextlib, runpy
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
d = json.loads(buf.getvalue().strip().splitlines()[-1]); r = d["roofline"]
print("%-14s attn_waves=%-3d %9.1f emb/s %8.3f ms/step  attention %.3f ms  gemm %.3f ms" % (wl, w, d["value"], d["ms_per_step"], r["per_family"]["attention"]["ms_per_step"], r["per_family"]["gemm"]["ms_per_step"]))
PY
  done
done
done
cat $OUT/ab.log
