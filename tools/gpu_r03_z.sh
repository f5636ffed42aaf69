#!/bin/bash
# round 3, call Z: round-end numbers of the other workloads on the final tree + a tile-height sweep inside the tower (the earlier sweeps ran the GEMMs alone)
tag=${1:-r03z}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
(rocm-smi --showclocks --showpower 2>&1 | grep -v "^$" | head -20) > $out/box_state.txt
for wl in add_documents_stream add_documents_mixed vit_l14_chunked_fp8 vit_l14_mixed; do
  timeout 500 python bench.py --workload $wl > $out/bench_$wl.json 2> $out/bench_$wl.err
  python -c "
import json; d = json.loads(open('$out/bench_$wl.json').read().strip().splitlines()[-1])
print('$wl', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'cpu', d.get('cpu_baseline', {}).get('value'), d.get('bf16_twin'))" 2>&1 | tee -a $out/workloads.txt
done
timeout 300 python tools/latency_bench.py 2>/dev/null | tee $out/latency.txt
for mt in 0 4 5 6; do
  MQ_GEMM_MT=$mt timeout 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('vit_b32_image gemm_mt=$mt  %9.1f emb/s %8.3f ms/step  gemm frac %.3f' % (d['value'], d['ms_per_step'], r['frac']))" | tee -a $out/mt_sweep_in_tower.txt
done
timeout 400 python tools/coalesce_bench.py --windows 0,1000 2>/dev/null | tee $out/coalesce_bench.txt
