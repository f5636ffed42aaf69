#!/bin/bash
# round 2, GPU call C: skinny (small-M) path — parity tests, single-request latency with and without it, kernel-level trace of the latency run
TAG=${1:-r02c}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_small_m_gpu.py tests/test_towers_gpu.py tests/test_edge_cases_gpu.py tests/test_gemm_variants_gpu.py \
  "tests/test_gpu_tokenizers.py::test_device_clip_bpe_equals_host" "tests/test_s2_inference_gpu.py::test_hf_xlm_roberta_from_disk" \
  -m gpu -q -s -x -p no:cacheprovider > $OUT/pytest_small.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_small.log
grep -E "passed|failed|rc=|skinny|Error|assert" $OUT/pytest_small.log | tail -20
python tools/latency_bench.py > $OUT/latency_small.txt 2>&1; tail -14 $OUT/latency_small.txt
MQ_SMALL_M=0 python tools/latency_bench.py > $OUT/latency_tiled.txt 2>&1; tail -14 $OUT/latency_tiled.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_latency -o lat -- python $GRAFT_REPO_ROOT/tools/latency_bench.py --only "ViT-B-32" > $OUT/prof_latency.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $OUT/prof_latency/*/*kernel_stats.csv $OUT/prof_latency/*kernel_stats.csv 2>/dev/null | head -1); echo "stats: $f"; head -25 "$f" | cut -c1-220
