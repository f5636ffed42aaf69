#!/bin/bash
# round 2, GPU call H: where do concurrent image calls serialise? (phase timing, 1 / 2 / 4 callers); ingest host profile
TAG=${1:-r02h}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for t in 1 2 4; do python tools/e2e_profile.py --threads $t >> $OUT/e2e_phases.txt 2>&1; done; grep -v amdgpu.ids $OUT/e2e_phases.txt
python - > $OUT/ingest_profile.txt 2>&1 <<'PY'
import cProfile, pstats, sys, os, io
sys.argv = ["bench.py", "--workload", "add_documents_mixed", "--steps", "20", "--warmup", "3"]
import runpy
pr = cProfile.Profile(); pr.enable()
try:
    runpy.run_path("bench.py", run_name="__main__")
finally:
    pr.disable(); s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40); print(s.getvalue())
PY
grep -v "^$" $OUT/ingest_profile.txt | cut -c1-190 | head -60
