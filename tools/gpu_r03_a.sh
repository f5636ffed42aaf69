#!/bin/bash
# round 3, call A: the two-accumulator-set GEMM (gemm_pp.hip): parity, then interleaved within-process A/B at the towers' shapes,
# then the headline with and without it; the side-car knob's parity + A/B left over from round 2.   usage: tools/gpu_r03_a.sh <tag>
tag=${1:-r03a}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gemm_variants_gpu.py -q -m gpu -x -k "two_accumulator" 2>&1 | tail -15 | tee $out/pp_pytest.txt
timeout 400 python tools/gemm_bench.py --ab "base:gemm_pp=0;pp2:gemm_pp=2,gemm_pp_pps=2;pp4:gemm_pp=2,gemm_pp_pps=4" --rounds 5 --iters 20 --only "b32|l14|4096" 2>&1 | grep -v amdgpu.ids | tee $out/pp_gemm_ab.txt
for pp in 0 1 2; do
  MQ_GEMM_PP=$pp timeout 200 python bench.py --steps 30 --warmup 8 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('gemm_pp=$pp  %9.1f emb/s %8.3f ms/step  gemm %6.1f TF (frac %.3f) avg launch %.1f us  fam %s' % (d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['avg_launch_us'], {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items()}))
" 2>&1 | tee -a $out/pp_headline_ab.txt
done
for pp in 0 1; do for rb in 0 1; do
  MQ_GEMM_PP=$pp MQ_RESIDUAL_BF16=$rb timeout 200 python bench.py --steps 30 --warmup 8 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('gemm_pp=$pp residual_bf16=$rb  %9.1f emb/s %8.3f ms/step  gemm %6.1f TF (frac %.3f)  fam %s' % (d['value'], d['ms_per_step'], r['achieved'], r['frac'], {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items()}))
" 2>&1 | tee -a $out/pp_headline_ab.txt
done; done
MARQO_AMD_PREPROCESS_SIDECAR=1 timeout 200 python -m pytest tests/test_s2_inference_gpu.py -q -m gpu -k "sidecar" 2>&1 | tail -3 | tee $out/sidecar_pytest.txt
for side in 0 1; do MARQO_AMD_PREPROCESS_SIDECAR=$side timeout 100 python tools/e2e_profile.py 2>&1 | grep "====" | sed "s/^/sidecar=$side /" | tee -a $out/sidecar_ab.log; done
