#!/bin/bash
# SigLIP towers on the GPU: parity tests + bench lines
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_towers_gpu.py tests/test_s2_inference_gpu.py tests/test_kernels_gpu.py -m gpu -q -x > gpurun_out/siglip_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/siglip_tests.log; tail -4 gpurun_out/siglip_tests.log
: > gpurun_out/siglip_bench.jsonl
for spec in "siglip_b16_image bf16" "siglip_b16_image fp8" "siglip_l16_384_image bf16" "siglip_b16_text bf16"; do
  set -- $spec
  extra=""; [ "$1 $2" = "siglip_b16_image bf16" ] && extra="--cpu-seconds 10" || extra="--no-cpu-baseline"
  timeout 300 python bench.py --workload $1 --precision $2 --steps 10 --warmup 3 $extra >> gpurun_out/siglip_bench.jsonl 2> gpurun_out/siglip_bench_$1_$2.err || tail -5 gpurun_out/siglip_bench_$1_$2.err
done
cat gpurun_out/siglip_bench.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['config']['workload'][:70], d['dtype'], d['value'], 'emb/s', d['ms_per_step'], 'ms', 'gemm', r['achieved'], 'TF', {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items()}, d.get('cpu_baseline'), d.get('cosine_delta_vs_cpu'))
"
