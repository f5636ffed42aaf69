#!/bin/bash
# native batch staging of Pillow images (_mq_stage.gather_rgbx) vs the per-image pyarrow export: tools/e2e_profile.py --threads {1,4}, two alternating rounds.
# usage: tools/gpu_r02_ag.sh <tag>
tag=${1:-r02ag}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_preprocess_gpu.py tests/test_image_modes.py -x -q -m gpu 2>&1 | tail -3 > $out/pytest_subset.txt; cat $out/pytest_subset.txt
for round in 1 2; do for ns in 0 1; do for th in 1 4; do
  MARQO_AMD_NATIVE_STAGE=$ns timeout 100 python tools/e2e_profile.py --threads $th 2>&1 | grep "====" | sed "s/^/native_stage=$ns /" | tee -a $out/ab.log
done; done; done
