#!/bin/bash
# GEMM memory-side diagnostic: the tower GEMM shapes with real operand addresses vs aliased ones (tools/probes/build_gemm_alias.sh),
# plus the s_memtime tick-rate calibration
mkdir -p gpurun_out
./tools/probes/memtime_calib > gpurun_out/memtime_calib.txt 2>&1
cat gpurun_out/memtime_calib.txt
for v in real alias2 alias1; do
  lib=marqo_amd/lib/libmarqo_hip.so
  [ $v != real ] && lib=tools/probes/libmarqo_hip_$v.so
  echo "== $v" >> gpurun_out/gemm_alias.txt
  MARQO_AMD_LIB=$PWD/$lib timeout 300 python tools/gemm_bench.py --iters 30 --only 'b32|l14|4096' >> gpurun_out/gemm_alias.txt 2>&1
done
cat gpurun_out/gemm_alias.txt
