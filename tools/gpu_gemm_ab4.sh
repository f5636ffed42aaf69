#!/bin/bash
TAG=${1:-gemm4}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_variants_gpu.py -q -x -k "cu_sized" 2>&1 | tail -15 > $OUT/ab.log
echo "== baseline" >> $OUT/ab.log
python tools/gemm_bench.py --iters 30 2>&1 | grep -v amdgpu.ids >> $OUT/ab.log
for big in 4 6 8; do
  echo "== BIG=$big" >> $OUT/ab.log
  MQ_GEMM_BIG=$big python tools/gemm_bench.py --iters 30 2>&1 | grep -v amdgpu.ids >> $OUT/ab.log
done
cat $OUT/ab.log
