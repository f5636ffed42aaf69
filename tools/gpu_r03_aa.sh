#!/bin/bash
# round 3, call AA: what costs the merged engine call its time under 16 request threads — per-thread streams (wait_stream / events at every call) and the interpreter's switch interval
tag=${1:-r03aa}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $out/smoke.txt
for ts in 1 0; do for si in 0 0.0002; do
  echo "== MARQO_AMD_THREAD_STREAMS=$ts switch-interval=$si" | tee -a $out/coalesce_contention_ab.txt
  MARQO_AMD_THREAD_STREAMS=$ts timeout 300 python tools/coalesce_bench.py --only ViT-B-32 --windows 0,1000 --switch-interval $si 2>/dev/null | tee -a $out/coalesce_contention_ab.txt
done; done
