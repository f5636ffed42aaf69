#!/bin/bash
TAG=${1:-gemm3}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_variants_gpu.py -q -x 2>&1 | tail -15 > $OUT/ab.log
for ring in 0 1; do
  echo "== RING=$ring MT=auto" >> $OUT/ab.log
  MQ_GEMM_RING=$ring python tools/gemm_bench.py --iters 30 2>&1 | grep -v amdgpu.ids >> $OUT/ab.log
  for mt in 4 5 6; do
    echo "== RING=$ring MT=$mt" >> $OUT/ab.log
    MQ_GEMM_RING=$ring MQ_GEMM_MT=$mt python tools/gemm_bench.py --iters 30 --only "4096|8192|fc1|fc2" 2>&1 | grep -v amdgpu.ids >> $OUT/ab.log
  done
done
for ring in 0 1; do
  echo "== bench RING=$ring" >> $OUT/ab.log
  MQ_GEMM_RING=$ring python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> $OUT/ab.log
done
cat $OUT/ab.log
