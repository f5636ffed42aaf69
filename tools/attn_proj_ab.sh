#!/bin/bash
# headline tower with / without the one-launch attention half (MQ_ATTN_PROJ = fewest images from which it is used, 0 = never) at several batch sizes, interleaved
# bash tools/attn_proj_ab.sh "<batch> ..."   (on the GPU box; prints embeddings/s, ms per step, per-family ms)
for b in ${1:-256 128 96 64}; do for rep in 1 2; do for v in 0 1; do
  MQ_ATTN_PROJ=$v python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pf=d['roofline']['per_family']
print('batch $b MQ_ATTN_PROJ=$v', d['value'], d['ms_per_step'], {k:(round(x['ms_per_step'],3), x['launches_per_step']) for k,x in pf.items() if k in ('gemm','layernorm','attention')})"
done; done; done
