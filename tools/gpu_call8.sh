#!/bin/bash
TAG=${1:-c8}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fp8_gpu.py -q -x -s -k "bert or rowquant" 2>&1 | grep -E "passed|failed|1-cos|Error|error" | tail -12 > $OUT/tests.log
cat $OUT/tests.log
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-34s %9.1f emb/s %8.3f ms/step  gemm %6.1f TF  fam %s' % ('$1', d['value'], d['ms_per_step'], r['achieved'], {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items()}))
"; }
for p in bf16 fp8; do
  timeout 300 python bench.py --workload bert_base_77 --precision $p --steps 20 --warmup 5 --no-cpu-baseline 2>$OUT/err.txt | line "bert_base_77 $p" >> $OUT/ab.log 2>&1 || tail -5 $OUT/err.txt >> $OUT/ab.log
done
cat $OUT/ab.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $OUT/pytest_gpu.log; cat $OUT/pytest_gpu.log
