#!/bin/bash
TAG=${1:-gemm6}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
AB="base:gemm_cgroup=0,gemm_wide=0,gemm_persist=0;cg8:gemm_cgroup=8,gemm_wide=0,gemm_persist=0;wide:gemm_cgroup=0,gemm_wide=1,gemm_persist=0;pers:gemm_cgroup=0,gemm_wide=0,gemm_persist=1;pw:gemm_cgroup=0,gemm_wide=1,gemm_persist=1;all:gemm_cgroup=8,gemm_wide=1,gemm_persist=1"
python tools/gemm_bench.py --iters 20 --rounds 7 --only "b32|l14|4096" --ab "$AB" 2>&1 | grep -v amdgpu.ids > $OUT/ab.log
cat $OUT/ab.log
timeout 600 python -m pytest tests/test_gpu_tokenizers.py tests/test_gemm_variants_gpu.py -q -m gpu 2>&1 | tail -8 > $OUT/tests.log
cat $OUT/tests.log
