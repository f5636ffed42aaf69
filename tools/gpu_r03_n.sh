#!/bin/bash
# round 3, call N: XCD-banded block order of the row-wise kernels (LayerNorm, attention follow the GEMMs' row bands): A/B on the workloads
tag=${1:-r03n}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for rnd in 1 2; do for wl in vit_b32_image clip_text_b32 vit_l14_image; do for band in 0 1; do
  MQ_XCD_BAND=$band timeout 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$wl xcd_band=$band  %9.1f emb/s %8.3f ms/step  gemm frac %.3f  fam %s' % (d['value'], d['ms_per_step'], r['frac'], {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items()}))
" 2>&1 | tee -a $out/xcd_band_ab.txt
done; done; done
timeout 600 python -m pytest tests/test_fp8_gpu.py tests/test_kernels_gpu.py tests/test_towers_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee $out/pytest_subset.txt
