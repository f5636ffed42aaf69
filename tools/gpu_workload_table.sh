#!/bin/bash
# The workload table of DESIGN.md §6: one bench line per (workload, precision), 20 steps each.  usage: tools/gpu_workload_table.sh <tag>
TAG=${1:-wl}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
: > $OUT/wl.log
for spec in "vit_b32_image bf16" "vit_b32_image fp8" "vit_l14_image bf16" "vit_l14_image fp8" "clip_text_b32 bf16" "clip_text_b32 fp8" \
            "clip_text_l14 bf16" "clip_text_l14 fp8" "bert_base_77 bf16" "bert_base_77 fp8" "vit_l14_mixed bf16" "vit_l14_mixed fp8" \
            "siglip_b16_image bf16" "siglip_b16_image fp8" "siglip_b16_text bf16" "siglip_l16_384_image bf16" "vit_h14_image bf16" "vit_h14_image fp8" "vit_bigg14_image bf16"; do
  set -- $spec
  timeout 200 python bench.py --workload $1 --precision $2 --steps 20 --warmup 5 --no-cpu-baseline 2>$OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('%-22s %-4s %9.1f emb/s %8.3f ms/step  e2e %6.1f TF  gemm %6.1f TF (frac %.3f)  fam %s' % ('$1', '$2', d['value'], d['ms_per_step'], d['e2e_tflops'], r['achieved'], r['frac'], {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items()}))
" >> $OUT/wl.log 2>&1 || tail -3 $OUT/err.txt >> $OUT/wl.log
done
cat $OUT/wl.log
