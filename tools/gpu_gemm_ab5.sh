#!/bin/bash
# A/B of the round-1b GEMM knobs: L2-blocked tile order (cgroup), widened bf16 stores (wide), persistent tile loop (persist)
TAG=${1:-gemm5}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_variants_gpu.py tests/test_kernels_gpu.py -q -x 2>&1 | tail -8 > $OUT/ab.log
run() { echo "== $1" >> $OUT/ab.log; env $2 python tools/gemm_bench.py --iters 30 --only "b32|l14|4096" 2>&1 | grep -v amdgpu.ids >> $OUT/ab.log; }
run "base (cgroup=0 wide=0 persist=0)" "MQ_GEMM_CGROUP=0 MQ_GEMM_WIDE=0 MQ_GEMM_PERSIST=0"
run "cgroup=8" "MQ_GEMM_CGROUP=8 MQ_GEMM_WIDE=0 MQ_GEMM_PERSIST=0"
run "cgroup=4" "MQ_GEMM_CGROUP=4 MQ_GEMM_WIDE=0 MQ_GEMM_PERSIST=0"
run "wide=1" "MQ_GEMM_CGROUP=0 MQ_GEMM_WIDE=1 MQ_GEMM_PERSIST=0"
run "persist=1" "MQ_GEMM_CGROUP=0 MQ_GEMM_WIDE=0 MQ_GEMM_PERSIST=1"
run "cgroup=8 wide=1" "MQ_GEMM_CGROUP=8 MQ_GEMM_WIDE=1 MQ_GEMM_PERSIST=0"
run "cgroup=8 wide=1 persist=1" "MQ_GEMM_CGROUP=8 MQ_GEMM_WIDE=1 MQ_GEMM_PERSIST=1"
run "base again" "MQ_GEMM_CGROUP=0 MQ_GEMM_WIDE=0 MQ_GEMM_PERSIST=0"
cat $OUT/ab.log
