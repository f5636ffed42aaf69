#!/bin/bash
# A/B of GEMM variants + PMC counters.  usage: tools/gpu_gemm_ab.sh <tag>
TAG=${1:-gemm}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
echo "== correctness (SPEC=1)" > $OUT/ab.log
MQ_GEMM_SPEC=1 timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_towers_gpu.py -q -x 2>&1 | tail -3 >> $OUT/ab.log
for spec in 0 1; do
  echo "== SPEC=$spec MT=auto" >> $OUT/ab.log
  MQ_GEMM_SPEC=$spec python tools/gemm_bench.py --iters 30 2>&1 | grep -v amdgpu.ids >> $OUT/ab.log
  for mt in 4 6; do
    echo "== SPEC=$spec MT=$mt" >> $OUT/ab.log
    MQ_GEMM_SPEC=$spec MQ_GEMM_MT=$mt python tools/gemm_bench.py --iters 30 --only "^3" 2>&1 | grep -v amdgpu.ids >> $OUT/ab.log
  done
done
for spec in 0 1; do
  echo "== bench SPEC=$spec" >> $OUT/ab.log
  MQ_GEMM_SPEC=$spec python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>&1 | grep -v amdgpu.ids | cut -c1-400 >> $OUT/ab.log
done
cd /tmp
for spec in 0 1; do
  MQ_GEMM_SPEC=$spec timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
     --kernel-trace --output-format csv -d $OUT/pmc_spec$spec -o pmc -- python $REPO/tools/gemm_bench.py --iters 3 --only "b32 fc1|b32 fc2|4096" > $OUT/pmc_spec$spec.log 2>&1
done
cd $REPO
python tools/pmc_summary.py $OUT/pmc_spec0 $OUT/pmc_spec1 >> $OUT/ab.log 2>&1
cat $OUT/ab.log
