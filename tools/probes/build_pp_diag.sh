#!/bin/bash
# Ablation builds of csrc/gemm_pp.hip (timing only — results are wrong by construction): tools/probes/libmarqo_hip_ppdiag<N>.so with
# -DMQ_PP_DIAG=<N>, N a bit set: 1 no global->LDS traffic in the k-steps, 2 no fragment reads, 4 no barrier, 8 no counted vmcnt wait,
# 16 no interleaved epilogue.  Every other object is reused from the regular build (csrc/.obj).   usage: build_pp_diag.sh 1 2 3 ...
set -e
cd "$(dirname "$0")/../.."
python -c "from marqo_amd import _lib; _lib.build()" > /dev/null
for mode in "$@"; do
  ( O=/tmp/ppdiag_$mode.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DMQ_PP_DIAG=$mode -c marqo_amd/csrc/gemm_pp.hip -o $O
    OBJS=$(ls marqo_amd/csrc/.obj/*.o | grep -v gemm_pp.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $O -o tools/probes/libmarqo_hip_ppdiag$mode.so
    echo built tools/probes/libmarqo_hip_ppdiag$mode.so ) &
done
wait
