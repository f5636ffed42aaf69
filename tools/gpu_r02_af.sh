#!/bin/bash
# after moving the request path's host bookkeeping from torch CPU ops to NumPy: text / tokenizer / loader tests, the ingest phases, the ingest bench line,
# the default bench line (e2e_vectorise + also).  usage: tools/gpu_r02_af.sh <tag>
tag=${1:-r02af}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_towers_gpu.py tests/test_gpu_tokenizers.py tests/test_s2_inference_gpu.py tests/test_small_m_gpu.py tests/test_configs_gpu.py \
    tests/test_nccl_gpu.py tests/test_ref_parity_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu 2>&1 | tail -5 > $out/pytest_subset.txt
cat $out/pytest_subset.txt
timeout 120 python tools/ingest_profile.py > $out/ingest_phases.txt 2>&1; grep "====" $out/ingest_phases.txt
timeout 120 python bench.py --workload add_documents_mixed > $out/bench_ingest.json 2> $out/err_ingest.log; cut -c1-160 $out/bench_ingest.json
timeout 300 python bench.py > $out/bench_default.json 2> $out/err_default.log; cut -c1-200 $out/bench_default.json
timeout 120 python tools/latency_bench.py > $out/latency.txt 2>&1; tail -12 $out/latency.txt
