#!/bin/bash
# Diagnostic builds of the library with -DMQ_GEMM_DIAG=1/2/3 (gemm_nt_kernel without its global->LDS traffic / without its MFMAs / without its
# epilogue): tools/probes/libmarqo_hip_diag{1,2,3}.so.  Timing only — results are wrong by construction.  Bounds what each resource costs.
set -e
cd "$(dirname "$0")/../.."
for mode in 1 2 3; do
  OBJ=marqo_amd/csrc/.obj_diag$mode
  mkdir -p $OBJ
  for f in marqo_amd/csrc/*.hip; do
    b=$(basename $f .hip)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DMQ_GEMM_DIAG=$mode -c $f -o $OBJ/$b.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o tools/probes/libmarqo_hip_diag$mode.so
  echo built tools/probes/libmarqo_hip_diag$mode.so
done
