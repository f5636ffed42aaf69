#!/bin/bash
# round 2, GPU call Z: chaining chip-filling tower calls of concurrent callers on the GPU (MARQO_AMD_CHAIN_LARGE_CALLS) — e2e with 1 / 2 / 4 callers,
# chain on / off, default (torch-ops) boundary and ctypes; thread-safety tests
TAG=${1:-r02z}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
timeout 300 python -m pytest tests/test_edge_cases_gpu.py tests/test_ingest.py -m gpu -q -p no:cacheprovider > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_sel.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" $OUT/pytest_sel.log | tail -5
for rep in 1 2; do
for chain in 1 0; do
  for t in 1 2 4; do
    env MARQO_AMD_CHAIN_LARGE_CALLS=$chain python tools/e2e_profile.py --threads $t 2>&1 | grep "====" | sed "s/^/chain=$chain /" >> $OUT/ab.log
  done
  env MARQO_AMD_CHAIN_LARGE_CALLS=$chain MARQO_AMD_BOUNDARY=ctypes python tools/e2e_profile.py --threads 4 2>&1 | grep "====" | sed "s/^/chain=$chain ctypes /" >> $OUT/ab.log
done
done
cat $OUT/ab.log
