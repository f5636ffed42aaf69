#!/bin/bash
# round 2, GPU call O: attention kernels at 5 waves per SIMD (92 VGPRs, no spill) against the default build (100 VGPRs -> 4 waves per SIMD)
TAG=${1:-r02o}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
for r in 1 2; do
for wl in vit_b32_image clip_text_b32 vit_l14_image bert_base_77; do
  for lib in default occ5; do
    if [ $lib = occ5 ]; then export MARQO_AMD_LIB=$PWD/tools/probes/libmarqo_hip_attn_occ5.so; else unset MARQO_AMD_LIB; fi
    python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>$OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-14s %-8s %9.1f emb/s %8.3f ms/step  attention %.3f ms  gemm %.3f ms' % ('$wl', '$lib', d['value'], d['ms_per_step'], r['per_family']['attention']['ms_per_step'], r['per_family']['gemm']['ms_per_step']))
" >> $OUT/ab.log 2>&1 || tail -3 $OUT/err.txt >> $OUT/ab.log
  done
done
done
unset MARQO_AMD_LIB
cat $OUT/ab.log
