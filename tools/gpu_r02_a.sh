#!/bin/bash
# round 2, GPU call A: full GPU suite (incl. the reference-parity, RCCL and full-depth tests) + the default bench line + ingest + fp8 policy lines
TAG=${1:-r02a}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|error|rc=|1-cos|ViT-L|CLIP text|BERT|cfg 5" $OUT/pytest_gpu.log | tail -40
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -c 3000 $OUT/bench_default.json; tail -5 $OUT/bench_default.err
timeout 300 python bench.py --workload add_documents_mixed --steps 10 --warmup 2 > $OUT/bench_ingest.json 2> $OUT/bench_ingest.err; tail -c 1200 $OUT/bench_ingest.json; tail -3 $OUT/bench_ingest.err
for spec in "vit_l14_image bf16 7e-4" "vit_l14_image fp8 7e-4" "vit_l14_image fp8 1" "vit_b32_image fp8 7e-4" "vit_b32_image fp8 1"; do
  set -- $spec
  MARQO_AMD_FP8_BUDGET=$3 timeout 300 python bench.py --workload $1 --precision $2 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>>$OUT/wl.err | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-14s %-4s budget %-5s %9.1f emb/s %8.3f ms/step  gemm %6.1f TF (frac %.3f) policy %s' % ('$1', '$2', '$3', d['value'], d['ms_per_step'], r['achieved'], r['frac'], d['config'].get('fp8_policy')))
" >> $OUT/wl.log 2>&1
done
cat $OUT/wl.log; tail -5 $OUT/wl.err
