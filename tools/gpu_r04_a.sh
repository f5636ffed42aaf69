#!/bin/bash
# round 4, call A: pipelined GEMM main loop — bit identity vs the old loop, interleaved A/B at the tower shapes, headline with either loop
OUT=$PWD/gpurun_out/r04a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/gemm_pl_check.py > $OUT/gemm_pl_check.txt 2>&1; echo "rc=$?" >> $OUT/gemm_pl_check.txt
tail -30 $OUT/gemm_pl_check.txt
for pl in 0 1 0 1; do
  MQ_GEMM_PL=$pl timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras > $OUT/bench_pl$pl.json 2>$OUT/bench_pl$pl.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_pl$pl.json").read().strip().splitlines()[-1])
print("pl=$pl", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("cos_err_vs_cpu"))
PY
done
MQ_GEMM_PL=1 timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_towers_gpu.py tests/test_gemm_variants_gpu.py -m gpu -x -q 2>&1 | tail -5
