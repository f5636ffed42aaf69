#!/bin/bash
# round 3, call P: fp8 towers on the bf16 residual stream (parity tests, then fp8 workloads with the stream forced fp32 / chosen by the policy, bf16 beside them);
# row-group skinny GEMM tests once more after the test fix
tag=${1:-r03p}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_fp8_gpu.py tests/test_small_m_gpu.py tests/test_towers_gpu.py tests/test_configs_gpu.py -q -m gpu -x -s 2>&1 | grep -v "^$" | tail -40 | tee $out/pytest_subset.txt
for rnd in 1 2; do for wl in vit_l14_image vit_b32_image clip_text_b32; do
  for mode in bf16 fp8:fp32 fp8:auto; do
    prec=${mode%%:*}; stream=${mode##*:}; [ "$prec" = "bf16" ] && stream=auto
    MARQO_AMD_RESIDUAL_STREAM=$stream timeout 300 python bench.py --workload $wl --precision $prec --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; p = d['config'].get('fp8_policy')
pol = ' '.join('[split %d +%d mlp, %s stream, err %.1e]' % (q['fp8_first_layer'], q.get('fp8_mlp_extra', 0), q.get('residual_stream'), q['calibration_err_vs_bf16']) for q in p) if p else ''
print('$wl $prec stream=$stream  %9.1f emb/s %8.3f ms/step  gemm frac %.3f  %s' % (d['value'], d['ms_per_step'], r['frac'], pol))
" 2>&1 | tee -a $out/fp8_stream_ab.txt
  done
done; done
for stream in fp32 auto; do
  MARQO_AMD_RESIDUAL_STREAM=$stream timeout 400 python bench.py --workload vit_l14_chunked_fp8 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null > $out/bench_chunked_fp8_$stream.json
  python -c "
import json; d = json.loads(open('$out/bench_chunked_fp8_$stream.json').read().strip().splitlines()[-1])
print('chunked fp8 stream=$stream', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('bf16_twin'), d['config'].get('fp8_policy'))" 2>&1 | cut -c1-900 | tee -a $out/fp8_stream_ab.txt
done
