#!/bin/bash
# round 3, call O: the skinny GEMM in row groups (81-320 rows): parity tests, headline A/B, few-item latency A/B; where a merged engine call spends its time
tag=${1:-r03o}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_small_m_gpu.py tests/test_towers_gpu.py tests/test_coalesce.py tests/test_edge_cases_gpu.py -q -m gpu -x 2>&1 | tail -8 | tee $out/pytest_subset.txt
for rnd in 1 2; do for wl in vit_b32_image clip_text_b32; do for grp in 0 320; do
  MQ_SMALL_M_GROUPED=$grp timeout 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$wl small_m_grouped=$grp  %9.1f emb/s %8.3f ms/step  gemm frac %.3f  fam %s' % (d['value'], d['ms_per_step'], r['frac'], {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items()}))
" 2>&1 | tee -a $out/grouped_ab.txt
done; done; done
for grp in 0 320; do echo "== MQ_SMALL_M_GROUPED=$grp" | tee -a $out/latency_ab.txt; MQ_SMALL_M_GROUPED=$grp timeout 300 python tools/latency_bench.py --only ViT-B-32 2>/dev/null | tee -a $out/latency_ab.txt; MQ_SMALL_M_GROUPED=$grp timeout 300 python tools/latency_bench.py --only e5-base-v2 2>/dev/null | tee -a $out/latency_ab.txt; done
timeout 300 python tools/coalesce_profile.py --no-profile 2>/dev/null | tee $out/coalesce_profile.txt
timeout 300 python tools/coalesce_profile.py 2>/dev/null | cut -c1-200 | tee -a $out/coalesce_profile.txt
