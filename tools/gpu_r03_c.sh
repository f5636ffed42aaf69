#!/bin/bash
# round 3, call C: where do the k-step's non-MFMA cycles of gemm_pp go?  Ablation builds (tools/probes/build_pp_diag.sh), timing only.
tag=${1:-r03c}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for d in real 1 2 3 4 8 15 16 31; do
  if [ $d = real ]; then unset MARQO_AMD_LIB; else export MARQO_AMD_LIB=$PWD/tools/probes/libmarqo_hip_ppdiag$d.so; fi
  echo "== diag $d" | tee -a $out/pp_ablation.txt
  MQ_GEMM_PP=2 MQ_GEMM_PP_PPS=4 timeout 120 python tools/gemm_bench.py --iters 20 --only "b32 qkv|b32 fc1|l14 qkv|4096|8192" 2>&1 | grep -v "amdgpu.ids\|get_num_threads" | tee -a $out/pp_ablation.txt
done
unset MARQO_AMD_LIB
echo "== shipped kernel (gemm_pp off)" | tee -a $out/pp_ablation.txt
timeout 120 python tools/gemm_bench.py --iters 20 --only "b32 qkv|b32 fc1|l14 qkv|4096|8192" 2>&1 | grep -v "amdgpu.ids\|get_num_threads" | tee -a $out/pp_ablation.txt
