#!/bin/bash
# round 3, call S2: weight prefetch gated on the tower's weight bytes (off for the text towers) and extended to the fp8 LayerNorms
tag=${1:-r03t}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for rnd in 1 2; do for spec in vit_b32_image:bf16 clip_text_b32:bf16 vit_b32_image:fp8 vit_l14_image:fp8; do for pf in 0 1; do
  wl=${spec%%:*}; prec=${spec##*:}
  MQ_LN_PREFETCH=$pf timeout 300 python bench.py --workload $wl --precision $prec --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$wl $prec ln_prefetch=$pf  %9.1f emb/s %8.3f ms/step  gemm frac %.3f  fam %s' % (d['value'], d['ms_per_step'], r['frac'], {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items()}))
" 2>&1 | tee -a $out/ln_prefetch_ab2.txt
done; done; done
timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_towers_gpu.py -q -m gpu -x 2>&1 | tail -4 | tee $out/pytest_subset.txt
