#!/bin/bash
# round 2, GPU call E: search-path tests + latency; tiled GEMM: LDS-staged bias A/B, second-workgroup stagger sweep
TAG=${1:-r02e}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_small_m_gpu.py tests/test_towers_gpu.py tests/test_edge_cases_gpu.py tests/test_gemm_variants_gpu.py tests/test_ln_fold_gpu.py \
  tests/test_s2_inference_gpu.py tests/test_kernels_gpu.py tests/test_configs_gpu.py tests/test_ref_parity_gpu.py -m gpu -q -s -x -p no:cacheprovider > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_sel.log
grep -E "passed|failed|rc=|skinny|^FAILED|^ERROR|Error" $OUT/pytest_sel.log | tail -12
python tools/latency_bench.py > $OUT/latency.txt 2>&1; tail -14 $OUT/latency.txt
python tools/gemm_bench.py --only "b32 (qkv|out|fc1|fc2)$|l14 (qkv|fc1)" --iters 30 --rounds 5 --ab "ldsbias:gemm_lds_bias=1;global:gemm_lds_bias=0" > $OUT/gemm_lds_bias_ab.txt 2>&1; cat $OUT/gemm_lds_bias_ab.txt
python tools/gemm_bench.py --only "b32 (qkv|fc1)$|l14 (qkv|fc1)" --iters 30 --rounds 5 --ab "s0:gemm_stagger=0;s4:gemm_stagger=4;s8:gemm_stagger=8;s12:gemm_stagger=12;s16:gemm_stagger=16" > $OUT/gemm_stagger_ab.txt 2>&1; cat $OUT/gemm_stagger_ab.txt
