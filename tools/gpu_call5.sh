#!/bin/bash
TAG=${1:-c5}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ln_fold_gpu.py tests/test_configs_gpu.py -q -x -s 2>&1 | grep -v "^$" | tail -40 > $OUT/new_tests.log
cat $OUT/new_tests.log
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-28s %9.1f emb/s %8.3f ms/step  e2e %6.1f TF  gemm %6.1f TF (frac %.3f)  fam %s' % ('$1', d['value'], d['ms_per_step'], d['e2e_tflops'], r['achieved'], r['frac'], {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items()}))
"; }
for rep in 1 2; do
for f in 0 1; do
  MQ_LN_FOLD=$f timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>$OUT/err.txt | line "b32 ln_fold=$f" >> $OUT/ab.log 2>&1
done
done
for wl in clip_text_b32 vit_l14_image; do for f in 0 1; do
  MQ_LN_FOLD=$f timeout 300 python bench.py --workload $wl --steps 15 --warmup 3 --no-cpu-baseline 2>$OUT/err.txt | line "$wl ln_fold=$f" >> $OUT/ab.log 2>&1
done; done
cat $OUT/ab.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/pytest_gpu.log; cat $OUT/pytest_gpu.log
