#!/bin/bash
# round 3, call D: the 8-wave form of gemm_pp (two waves per SIMD): parity, then A/B against the shipped kernel and the 4-wave form
tag=${1:-r03d}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_variants_gpu.py -q -m gpu -x -k "two_accumulator" 2>&1 | tail -12 | tee $out/pp_pytest.txt
timeout 600 python tools/gemm_bench.py --ab "base:gemm_pp=0;pp8l:gemm_pp=2,gemm_pp_waves=8,gemm_pp_pps=2;pp8h:gemm_pp=2,gemm_pp_waves=8,gemm_pp_pps=4;pp4h:gemm_pp=2,gemm_pp_waves=4,gemm_pp_pps=4" --rounds 5 --iters 20 --only "b32|l14|4096" 2>&1 | grep -v "amdgpu.ids\|get_num_threads" | tee $out/pp8_gemm_ab.txt
