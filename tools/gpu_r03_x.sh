#!/bin/bash
# round 3, call X: granularity of the weight prefetch (one dword per 128 / 64 / 32 bytes): does a dword touch bring the whole 128-byte line?
tag=${1:-r03x}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for rnd in 1 2; do for wl in vit_b32_image vit_l14_image; do for pf in 0 1 2 3; do
  MQ_LN_PREFETCH=$pf timeout 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$wl ln_prefetch=$pf  %9.1f emb/s %8.3f ms/step  gemm frac %.3f  fam %s' % (d['value'], d['ms_per_step'], r['frac'], {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items() if k in ('gemm', 'layernorm', 'attention')}))
" 2>&1 | tee -a $out/ln_prefetch_granularity_ab.txt
done; done; done
