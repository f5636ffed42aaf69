#!/bin/bash
# Queued A/Bs that ran out of GPU budget in round 2: run first thing with a fresh budget.  usage: tools/gpu_next_ab.sh <tag>
#   1. 256-image synchronous calls as two 128-image pipeline stages (MARQO_AMD_IMAGE_PIPELINE_CHUNK=128) vs one stage (default 512)
#   2. a lone short query through the loaders: host tokeniser (default) vs the device route (MARQO_AMD_HOST_TOKENIZE_MAX_CHARS=-1)
#      (tools/latency_bench.py measures tower calls on ready ids; the ingest profile's text phase shows the tokeniser share)
tag=${1:-next}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for round in 1 2; do for chunk in 512 128; do for th in 1 4; do
  MARQO_AMD_IMAGE_PIPELINE_CHUNK=$chunk timeout 100 python tools/e2e_profile.py --threads $th 2>&1 | grep "====" | sed "s/^/pipeline_chunk=$chunk /" | tee -a $out/pipeline_chunk_ab.log
done; done; done
# 3. `.preprocess` side-car (MARQO_AMD_PREPROCESS_SIDECAR=1): first its parity (device-tensor route == uint8 route, bit for bit), then the rate
MARQO_AMD_PREPROCESS_SIDECAR=1 timeout 200 python -m pytest tests/test_s2_inference_gpu.py -q -m gpu -k "sidecar" 2>&1 | tail -3 | tee $out/sidecar_pytest.txt
for side in 0 1; do MARQO_AMD_PREPROCESS_SIDECAR=$side timeout 100 python tools/e2e_profile.py 2>&1 | grep "====" | sed "s/^/sidecar=$side /" | tee -a $out/sidecar_ab.log; done
timeout 120 python tools/ingest_profile.py > $out/ingest_phases.txt 2>&1; grep "====" $out/ingest_phases.txt
