#!/bin/bash
# round 2, GPU call G: RGBX staging / pipelined image calls — tests, e2e numbers + host profile; vendor-library GEMM calibration
TAG=${1:-r02g}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_preprocess_gpu.py tests/test_s2_inference_gpu.py tests/test_configs_gpu.py tests/test_ref_parity_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_sel.log
grep -E "passed|failed|rc=|^FAILED|^ERROR|Error" $OUT/pytest_sel.log | tail -12
python tools/e2e_profile.py > $OUT/e2e_profile.txt 2>&1; grep "====" $OUT/e2e_profile.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; python -c "
import json; d = json.load(open('$OUT/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac']); print(json.dumps(d.get('e2e_vectorise'), indent=1))"
python tools/blas_ref_bench.py > $OUT/blas_ref.txt 2>&1; cat $OUT/blas_ref.txt
python bench.py --workload add_documents_mixed --steps 10 --warmup 2 > $OUT/bench_ingest.json 2> $OUT/bench_ingest.err; python -c "
import json; d = json.load(open('$OUT/bench_ingest.json')); print('ingest', d['value'], d['ms_per_step'])"
