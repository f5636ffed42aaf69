#!/bin/bash
# Diagnostic builds of the library with -DMQ_GEMM_ALIAS=1 / 2 (gemm_nt_kernel fetches its operands from aliased tile origins:
# 1 = one tile for everybody, L1/L2-hot; 2 = 4 x 4 tiles, L2-hot): tools/probes/libmarqo_hip_alias{1,2}.so.  Timing only —
# results are wrong by construction.  Separates "the CU-side global->LDS path" from "the L2-miss / fabric side".
set -e
cd "$(dirname "$0")/../.."
for mode in 1 2; do
  OBJ=marqo_amd/csrc/.obj_alias$mode
  mkdir -p $OBJ
  for f in marqo_amd/csrc/*.hip; do
    b=$(basename $f .hip)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DMQ_GEMM_ALIAS=$mode -c $f -o $OBJ/$b.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o tools/probes/libmarqo_hip_alias$mode.so
  echo built tools/probes/libmarqo_hip_alias$mode.so
done
