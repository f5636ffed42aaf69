#!/bin/bash
# end-of-round check: the whole GPU suite, smoke(), the default bench line and the ingest line.  usage: tools/gpu_r02_ah.sh <tag>
tag=${1:-r02ah}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 420 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $out/pytest_gpu.txt; cat $out/pytest_gpu.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
timeout 120 python bench.py --workload add_documents_mixed > $out/bench_ingest.json 2> $out/err_ingest.log; cut -c1-160 $out/bench_ingest.json
timeout 200 python bench.py > $out/bench_default.json 2> $out/err_default.log; cut -c1-200 $out/bench_default.json
