#!/bin/bash
# round 3, call V: does a smaller per-call row count (working set inside the Infinity Cache) run faster per item on the text towers?
tag=${1:-r03v}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for spec in bert_base_77:1024 bert_base_77:512 bert_base_77:256 bert_base_77:128 clip_text_b32:1024 clip_text_b32:512 clip_text_b32:256 vit_b32_image:256 vit_b32_image:128 vit_b32_image:512 vit_l14_image:128 vit_l14_image:64; do
  wl=${spec%%:*}; b=${spec##*:}
  timeout 300 python bench.py --workload $wl --batch $b --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$wl batch=$b  %9.1f emb/s %8.3f ms/step  gemm frac %.3f  fam %s' % (d['value'], d['ms_per_step'], r['frac'], {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items()}))
" 2>&1 | tee -a $out/batch_sweep.txt
done
