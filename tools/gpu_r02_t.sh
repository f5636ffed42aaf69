#!/bin/bash
# round 2, GPU call T: does releasing the GIL inside the tower ops help or hurt concurrent callers?  e2e phase timing (tools/e2e_profile.py --threads)
# with the torch-ops boundary releasing / holding the GIL and with the ctypes boundary (which always releases it), 1 / 2 / 4 callers, one box
TAG=${1:-r02t}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
for rep in 1 2; do
for mode in release hold ctypes; do
  for t in 1 4; do
    case $mode in
      release) env MARQO_AMD_OPS_RELEASE_GIL=1 python tools/e2e_profile.py --threads $t 2>&1 | grep "====" | sed "s/^/$mode /" >> $OUT/ab.log ;;
      hold)    env MARQO_AMD_OPS_RELEASE_GIL=0 python tools/e2e_profile.py --threads $t 2>&1 | grep "====" | sed "s/^/$mode /" >> $OUT/ab.log ;;
      ctypes)  env MARQO_AMD_BOUNDARY=ctypes python tools/e2e_profile.py --threads $t 2>&1 | grep "====" | sed "s/^/$mode /" >> $OUT/ab.log ;;
    esac
  done
done
done
cat $OUT/ab.log
