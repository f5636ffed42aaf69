#!/bin/bash
# round 3, call B: gemm_pp wave-skew knob A/B (theory: the 4 lock-stepped waves queue at the CU's address path on every LDS-DMA piece) + the QUICKGELU mismatch diagnostic
tag=${1:-r03b}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 200 python tools/pp_diag.py 2>&1 | grep -v amdgpu.ids | tee $out/pp_diag.txt
timeout 500 python tools/gemm_bench.py --ab "base:gemm_pp=0;pp_s0:gemm_pp=2,gemm_pp_pps=4,gemm_pp_skew=0;pp_s1:gemm_pp=2,gemm_pp_pps=4,gemm_pp_skew=1;pp_s2:gemm_pp=2,gemm_pp_pps=4,gemm_pp_skew=2;pp_s3:gemm_pp=2,gemm_pp_pps=4,gemm_pp_skew=3;pp_s5:gemm_pp=2,gemm_pp_pps=4,gemm_pp_skew=5" --rounds 4 --iters 20 --only "b32 qkv|b32 fc1|b32 fc2_16|l14 qkv|l14 fc1|4096" 2>&1 | grep -v amdgpu.ids | tee $out/pp_skew_ab.txt
