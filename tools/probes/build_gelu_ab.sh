#!/bin/bash
# A/B build of the library with -DMQ_GELU_AS: the Abramowitz-Stegun (v_rcp + v_exp) erf-GELU instead of the polynomial form of
# csrc/common.h::gelu_erf2 -> tools/probes/libmarqo_hip_gelu_as.so (use: MARQO_AMD_LIB=... python tools/gemm_bench.py)
set -e
cd "$(dirname "$0")/../.."
OBJ=marqo_amd/csrc/.obj_gelu_as
mkdir -p $OBJ
for f in marqo_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DMQ_GELU_AS -c $f -o $OBJ/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o tools/probes/libmarqo_hip_gelu_as.so
echo built tools/probes/libmarqo_hip_gelu_as.so
