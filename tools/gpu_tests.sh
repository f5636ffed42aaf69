#!/bin/bash
# gpurun helper: GPU parity suite + optional extra commands.  usage: tools/gpu_tests.sh <tag> [extra shell command]
TAG=${1:-t}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
if [ -n "$2" ]; then bash -c "$2" > $OUT/extra.log 2>&1; tail -40 $OUT/extra.log; fi
