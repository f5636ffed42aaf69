#!/bin/bash
# round 2, GPU call F: remaining test files; tiled GEMM resource diagnostics (no-load / no-MFMA / no-epilogue builds)
TAG=${1:-r02f}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_s2_inference_gpu.py tests/test_kernels_gpu.py tests/test_configs_gpu.py tests/test_ref_parity_gpu.py tests/test_fp8_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_sel.log
grep -E "passed|failed|rc=|^FAILED|^ERROR|Error" $OUT/pytest_sel.log | tail -12
for v in real diag1 diag2 diag3; do
  echo "== $v" >> $OUT/gemm_diag.txt
  if [ $v = real ]; then python tools/gemm_bench.py --only "b32 (qkv|out|fc1|fc2)$|l14 |4096" --iters 30 >> $OUT/gemm_diag.txt 2>&1
  else MARQO_AMD_LIB=tools/probes/libmarqo_hip_$v.so python tools/gemm_bench.py --only "b32 (qkv|out|fc1|fc2)$|l14 |4096" --iters 30 >> $OUT/gemm_diag.txt 2>&1; fi
done
cat $OUT/gemm_diag.txt
