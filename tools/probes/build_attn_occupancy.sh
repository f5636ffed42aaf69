#!/bin/bash
# A/B build of the library with -DMQ_ATTN_WAVES_PER_EU=<n> (default 5): the attention kernels compiled for n waves per SIMD
# (amdgpu_waves_per_eu) -> tools/probes/libmarqo_hip_attn_occ<n>.so (use: MARQO_AMD_LIB=... python bench.py ...)
set -e
cd "$(dirname "$0")/../.."
N=${1:-5}
OBJ=marqo_amd/csrc/.obj_attn_occ
mkdir -p $OBJ
for f in marqo_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DMQ_ATTN_WAVES_PER_EU=$N -c $f -o $OBJ/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o tools/probes/libmarqo_hip_attn_occ$N.so
echo built tools/probes/libmarqo_hip_attn_occ$N.so
