#!/bin/bash
# round 3, call R: LayerNorm kernels prefetch the next GEMMs' weights into the Infinity Cache (MQ_LN_PREFETCH): A/B on the workloads + tower tests
tag=${1:-r03r}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for rnd in 1 2; do for wl in vit_b32_image clip_text_b32 vit_l14_image bert_base_77; do for pf in 0 1; do
  MQ_LN_PREFETCH=$pf timeout 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$wl ln_prefetch=$pf  %9.1f emb/s %8.3f ms/step  gemm frac %.3f  fam %s' % (d['value'], d['ms_per_step'], r['frac'], {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items()}))
" 2>&1 | tee -a $out/ln_prefetch_ab.txt
done; done; done
timeout 900 python -m pytest tests/test_towers_gpu.py tests/test_kernels_gpu.py tests/test_small_m_gpu.py -q -m gpu -x 2>&1 | tail -4 | tee $out/pytest_subset.txt
