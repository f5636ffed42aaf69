#!/bin/bash
# Developer aid (CPU only): compile ONE epilogue instantiation set of csrc/gemm_pp.hip and dump one kernel's ISA + resource usage.
#   tools/probes/pp_isa.sh <flags> <MT> <PPS>      e.g.  tools/probes/pp_isa.sh 1 4 2   ->  /tmp/probe/k.s
FL=${1:-1}; MT=${2:-4}; PPS=${3:-2}
mkdir -p /tmp/probe && cd /tmp/probe
python3 - <<PY
s=open('/root/repo/marqo_amd/csrc/gemm_pp.hip').read()
i=s.index('MQ_PP_INST(0);')
s=s[:i]+'MQ_PP_INST($FL);\n'
s=s.replace('#include "common.h"','#include "/root/repo/marqo_amd/csrc/common.h"')
open('ppdev.hip','w').write(s)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Rpass-analysis=kernel-resource-usage -c ppdev.hip -o ppdev.o -save-temps > ppdev.log 2>&1 || { grep -E "error" -A5 ppdev.log | head -40; exit 1; }
grep -E "Function Name|VGPRs:|AGPRs|Spill|Scratch" ppdev.log | sed 's/.*ppdev.hip:[0-9]*:[0-9]*: *//; s/\[-Rpass.*//' | paste - - - - - - | sed 's/Function Name: _ZN12_GLOBAL__N_114gemm_pp_kernelILi\([0-9]*\)ELi\([0-9]*\)ELi\([0-9]*\)E[^ \t]*/F\1 MT\2 PPS\3/'
S=ppdev-hip-amdgcn-amd-amdhsa-gfx950.s
WN=${4:-4}; awk -v pat="gemm_pp_kernelILi${FL}ELi${MT}ELi${PPS}ELi${WN}E" '$0 ~ "^_ZN.*"pat".*:" {f=1} f{print} f && /^\.Lfunc_end/{exit}' $S > k.s
echo "lines $(wc -l < k.s) mfma $(grep -c v_mfma k.s) glds $(grep -c global_load_lds k.s) dsread $(grep -c ds_read_b128 k.s) barrier $(grep -c s_barrier k.s) scratch $(grep -c scratch_ k.s) accread $(grep -c v_accvgpr_read k.s) accwrite $(grep -c v_accvgpr_write k.s) nop $(grep -c s_nop k.s)"
