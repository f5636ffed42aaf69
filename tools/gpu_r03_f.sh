#!/bin/bash
# round 3, call F: bf16-input LayerNorm forms under the bf16 residual stream (A/B), then the GPU suite with the stream on
tag=${1:-r03f}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for wl in vit_b32_image clip_text_b32 vit_l14_image; do for cfg in "0 0" "1 0" "1 1" "1 4"; do
  set -- $cfg
  MQ_RESIDUAL_BF16=$1 MQ_LN_BF16_WIDE=$2 timeout 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$wl residual_bf16=$1 ln_wide=$2  %9.1f emb/s %8.3f ms/step  gemm frac %.3f  fam %s' % (d['value'], d['ms_per_step'], r['frac'], {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items()}))
" 2>&1 | tee -a $out/residual_ln_ab.txt
done; done
MQ_RESIDUAL_BF16=1 timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_gemm_variants_gpu.py 2>&1 | tail -25 | tee $out/pytest_gpu_residual_bf16.txt
