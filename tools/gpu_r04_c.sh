#!/bin/bash
# round 4, call C: LayerNorm fold on the software-pipelined GEMM loop (persistent), A/B of fold x main loop on three towers
OUT=$PWD/gpurun_out/r04c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ln_fold_gpu.py tests/test_towers_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -30 > $OUT/pytest_fold.txt; tail -14 $OUT/pytest_fold.txt
one() {  # workload fold pl steps
  MQ_LN_FOLD=$2 MQ_GEMM_PL=$3 timeout 300 python bench.py --workload $1 --steps $4 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pf=d['roofline']['per_family']
print('$1 fold=$2 pl=$3', d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(round(v['ms_per_step'],3), v['launches_per_step']) for k,v in pf.items() if k in ('gemm','layernorm','attention')})"
}
for rep in 1 2; do
  one vit_b32_image 0 0 30; one vit_b32_image 1 0 30; one vit_b32_image 1 1 30; one vit_b32_image 0 1 30
done
for wl in vit_l14_image clip_text_b32 vit_l14_mixed; do
  one $wl 0 0 10; one $wl 1 0 10; one $wl 1 1 10; one $wl 0 1 10
done
one bert_base_77 0 0 10; one bert_base_77 0 1 10
