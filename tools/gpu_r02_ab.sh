#!/bin/bash
# round 2, GPU call AB: pack -> H2D pipelined in pieces (MARQO_AMD_PACK_CHUNKS) — preprocessing / loader tests, then e2e with 1 and 4 callers, chunks 1 vs 4
TAG=${1:-r02ab}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
timeout 400 python -m pytest tests/test_preprocess_gpu.py tests/test_s2_inference_gpu.py tests/test_configs_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_sel.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" $OUT/pytest_sel.log | tail -5
for rep in 1 2 3; do
for ch in 1 4; do
  for t in 1 4; do
    env MARQO_AMD_PACK_CHUNKS=$ch python tools/e2e_profile.py --threads $t 2>&1 | grep "====" | sed "s/^/chunks=$ch /" >> $OUT/ab.log
  done
done
done
cat $OUT/ab.log
