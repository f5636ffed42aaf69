#!/bin/bash
TAG=${1:-c9}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_tokenizers.py tests/test_s2_inference_gpu.py tests/test_configs_gpu.py -q -m gpu 2>&1 | tail -4 > $OUT/tests.log
cat $OUT/tests.log
timeout 600 python tools/tokenize_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/tok.log
cat $OUT/tok.log
