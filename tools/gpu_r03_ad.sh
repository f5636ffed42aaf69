#!/bin/bash
# round 3, call AD: several (sequence, head) items per attention workgroup for <= 64-token sequences: bit-identity test, then A/B on the workloads
tag=${1:-r03ad}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "attention" 2>&1 | tail -4 | tee $out/pytest_attention.txt
for rnd in 1 2; do for wl in vit_b32_image clip_text_b32; do for items in 1 2 3; do
  MQ_ATTN_ITEMS=$items timeout 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$wl attn_items=$items  %9.1f emb/s %8.3f ms/step  gemm frac %.3f  fam %s' % (d['value'], d['ms_per_step'], r['frac'], {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items() if k in ('gemm', 'layernorm', 'attention')}))
" 2>&1 | tee -a $out/attn_items_ab.txt
done; done; done
