#!/bin/bash
TAG=${1:-c7}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_gemm_variants_gpu.py tests/test_ln_fold_gpu.py -q -x 2>&1 | tail -8 > $OUT/tests.log
cat $OUT/tests.log
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-34s %9.1f emb/s %8.3f ms/step  gemm %6.1f TF  fam %s' % ('$1', d['value'], d['ms_per_step'], r['achieved'], {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items()}))
"; }
for rep in 1 2; do
for kn in "0 0 0" "1 8 2"; do
  set -- $kn
  for wl in vit_b32_image vit_l14_image clip_text_b32; do
    MQ_GEMM_PERSIST=$1 MQ_GEMM_CGROUP=$2 MQ_GEMM_WIDE=$3 timeout 300 python bench.py --workload $wl --precision fp8 --steps 20 --warmup 5 --no-cpu-baseline 2>$OUT/err.txt | line "$wl fp8 persist=$1 cg=$2 wide=$3" >> $OUT/ab.log 2>&1
  done
done
done
cat $OUT/ab.log
