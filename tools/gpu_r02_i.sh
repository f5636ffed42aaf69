#!/bin/bash
# round 2, GPU call I: evidence run — full GPU suite, rocprofv3 kernel trace + PMC passes of the bench command, default bench line, workload table,
# latency, ingest
TAG=${1:-r02m}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
bash tools/gpu_round_evidence.sh $TAG
python tools/latency_bench.py > $OUT/latency.txt 2>&1; tail -13 $OUT/latency.txt
python bench.py --workload add_documents_mixed --steps 20 --warmup 3 > $OUT/bench_ingest.json 2> $OUT/bench_ingest.err; python -c "
import json; d = json.load(open('$OUT/bench_ingest.json')); print('ingest', d['value'], d['ms_per_step'])"
python -c "
import json; d = json.load(open('$OUT/bench.json')); print(json.dumps(d.get('e2e_vectorise'), indent=1)); print(json.dumps(d['roofline'], indent=1)[:1500])"
