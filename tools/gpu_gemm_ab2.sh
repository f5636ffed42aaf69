#!/bin/bash
TAG=${1:-gemm2}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/ab.log
for spec in 0 1; do for mt in 2 4 5 6; do
  echo "== SPEC=$spec MT=$mt" >> $OUT/ab.log
  MQ_GEMM_SPEC=$spec MQ_GEMM_MT=$mt python tools/gemm_bench.py --iters 30 --only "4096|8192" 2>&1 | grep -v amdgpu.ids >> $OUT/ab.log
done; done
for gd in 2 4; do for mt in 4 6; do
  echo "== GD=$gd MT=$mt" >> $OUT/ab.log
  MQ_GEMM_GD=$gd MQ_GEMM_MT=$mt python tools/gemm_bench.py --iters 30 --only "4096|8192" 2>&1 | grep -v amdgpu.ids >> $OUT/ab.log
done; done
cat $OUT/ab.log
