#!/bin/bash
# One gpurun call: GPU parity tests, bench line, rocprofv3 kernel-trace stats.  Outputs under gpurun_out/.
# usage: tools/gpu_check.sh [tag]
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > $OUT/device.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/bench.err
REPO=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o vit_b32 -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/prof.err
cd $REPO

tail -3 $OUT/pytest_gpu.log; cat $OUT/bench.json; head -12 $OUT/kernel_stats.csv 2>/dev/null
python tools/rocpd_summary.py $(ls $OUT/prof/*results.db | head -1) $OUT/kernel_stats.csv 2>>$OUT/prof.err
head -8 $OUT/kernel_stats.csv | cut -c1-220
