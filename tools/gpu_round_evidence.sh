#!/bin/bash
# One call: full GPU suite + evidence for the headline workload + the workload table (bf16 / fp8).
TAG=${1:-r01c}
bash tools/gpu_evidence.sh $TAG
OUT=$PWD/gpurun_out/$TAG
for wl in vit_b32_image vit_l14_image clip_text_b32 clip_text_l14 bert_base_77 vit_l14_mixed; do
  for prec in bf16 fp8; do
    if [ $wl = bert_base_77 ] && [ $prec = fp8 ]; then continue; fi
    python bench.py --workload $wl --precision $prec --steps 20 --warmup 5 --no-cpu-baseline 2>$OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('%-14s %-4s %9.1f emb/s %8.3f ms/step  e2e %6.1f TF  gemm %6.1f TF (frac %.3f)  fam %s' % ('$wl', '$prec', d['value'], d['ms_per_step'], d['e2e_tflops'], r['achieved'], r['frac'], {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items()}))
" >> $OUT/wl.log 2>&1 || tail -3 $OUT/err.txt >> $OUT/wl.log
  done
done
cat $OUT/wl.log
