#!/bin/bash
# where does an add_documents_mixed step (128 strings + 128 PIL images through BulkVectoriser) spend its host time?  usage: tools/gpu_r02_ac.sh <tag>
tag=${1:-r02ac}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 240 python -m cProfile -o $out/ingest.prof bench.py --workload add_documents_mixed --steps 40 --warmup 5 > $out/bench_ingest.json 2> $out/err.log
python - <<PY > $out/ingest_profile.txt 2>&1
import pstats
for key in ("tottime", "cumtime"):
    print("=" * 30, key)
    pstats.Stats("$out/ingest.prof").strip_dirs().sort_stats(key).print_stats(45)
PY
rm -f $out/ingest.prof
cat $out/bench_ingest.json; head -70 $out/ingest_profile.txt | cut -c1-160
