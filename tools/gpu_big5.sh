#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gemm_variants_gpu.py -m gpu -q -k "cu_sized" > gpurun_out/big5_tests.log 2>&1; tail -3 gpurun_out/big5_tests.log
timeout 400 python tools/gemm_bench.py --iters 30 --rounds 5 --only 'b32|l14' --ab 'base:gemm_big=0;big5:gemm_big=5;big6:gemm_big=6' > gpurun_out/big5_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/big5_ab.txt
