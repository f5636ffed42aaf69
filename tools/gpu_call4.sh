#!/bin/bash
TAG=${1:-c4}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
AB="base:gemm_cgroup=0,gemm_wide=0,gemm_persist=0;w1:gemm_cgroup=8,gemm_wide=1,gemm_persist=1;w2:gemm_cgroup=8,gemm_wide=2,gemm_persist=1;w2np:gemm_cgroup=8,gemm_wide=2,gemm_persist=0"
python tools/gemm_bench.py --iters 20 --rounds 5 --only "b32|l14 (qkv|fc1)" --ab "$AB" 2>&1 | grep -v amdgpu.ids > $OUT/ab.log
cat $OUT/ab.log
bash tools/gpu_evidence.sh $TAG
