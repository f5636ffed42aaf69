#!/bin/bash
# Diagnostic build of the library with -DMQ_GEMM_TRACE (per-wave phase cycle sums in gemm_nt_kernel): tools/probes/libmarqo_hip_trace.so
set -e
cd "$(dirname "$0")/../.."
OBJ=marqo_amd/csrc/.obj_trace
mkdir -p $OBJ
for f in marqo_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DMQ_GEMM_TRACE -c $f -o $OBJ/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o tools/probes/libmarqo_hip_trace.so
echo built tools/probes/libmarqo_hip_trace.so
