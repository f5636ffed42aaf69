#!/bin/bash
# GPU check of the 128-wide-head attention path: parity tests + a short ViT-H/14 and ViT-bigG/14 bench
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_towers_gpu.py tests/test_fp8_gpu.py -m gpu -q -x > gpurun_out/wide_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/wide_tests.log
tail -5 gpurun_out/wide_tests.log
for w in vit_h14_image vit_bigg14_image; do
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/wide_bench_$w.json 2> gpurun_out/wide_bench_$w.err || tail -5 gpurun_out/wide_bench_$w.err
  cat gpurun_out/wide_bench_$w.json
done
timeout 300 python bench.py --workload vit_h14_image --precision fp8 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/wide_bench_h14_fp8.json 2> gpurun_out/wide_bench_h14_fp8.err || tail -5 gpurun_out/wide_bench_h14_fp8.err
cat gpurun_out/wide_bench_h14_fp8.json
