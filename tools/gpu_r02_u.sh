#!/bin/bash
# round 2, GPU call U: GEMM experiment knobs — counted vmcnt after an epilogue (gemm_vmcnt) and static priority for the second workgroup of a
# CU (gemm_prio): bit-identity test + interleaved within-process A/B at the tower shapes
TAG=${1:-r02u}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gemm_variants_gpu.py -m gpu -q -x -p no:cacheprovider -k "counted_vmcnt" > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_sel.log
grep -E "passed|failed|rc=|^FAILED|^ERROR|^E  " $OUT/pytest_sel.log | tail -6
python tools/gemm_bench.py --only "b32 (qkv|out|fc1|fc2)$|l14 (qkv|fc1)" --ab "base:gemm_vmcnt=0,gemm_prio=0;vmcnt:gemm_vmcnt=1,gemm_prio=0;prio:gemm_vmcnt=0,gemm_prio=1;both:gemm_vmcnt=1,gemm_prio=1" --rounds 7 > $OUT/ab.txt 2>&1
grep -v amdgpu.ids $OUT/ab.txt
