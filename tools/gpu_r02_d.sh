#!/bin/bash
# round 2, GPU call D: full GPU suite after the search-path rework (skinny GEMMs up to 272 rows, fused post-LN), latency, GEMM span trace
TAG=${1:-r02d}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|rc=|skinny|^FAILED|^ERROR" $OUT/pytest_gpu.log | tail -20
python tools/latency_bench.py > $OUT/latency.txt 2>&1; tail -14 $OUT/latency.txt
MARQO_AMD_LIB=tools/probes/libmarqo_hip_trace.so MQ_TRACE_VERBOSE=1 python tools/probes/gemm_trace.py > $OUT/gemm_trace.txt 2>&1; cat $OUT/gemm_trace.txt
