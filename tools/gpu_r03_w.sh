#!/bin/bash
# round 3, call W: post-LN (BERT family) bf16 stream: parity test, then A/B on the BERT workload (and the box's partition / clock state for the record)
tag=${1:-r03w}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
(rocm-smi --showmemorypartition --showcomputepartition --showclocks --showpower --showmaxpower 2>&1 | grep -v "^$" | head -40) > $out/box_state.txt
timeout 900 python -m pytest tests/test_towers_gpu.py -q -m gpu -x -s -k "post_ln_bf16_stream or row_selected or golden or bert" 2>&1 | grep -v "^$" | tail -22 | tee $out/pytest_subset.txt
for rnd in 1 2; do for stream in fp32 auto; do
  MARQO_AMD_RESIDUAL_STREAM=$stream timeout 300 python bench.py --workload bert_base_77 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('bert_base_77 stream=$stream  %9.1f emb/s %8.3f ms/step  gemm frac %.3f  fam %s' % (d['value'], d['ms_per_step'], r['frac'], {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items()}))
" 2>&1 | tee -a $out/post_ln_stream_ab.txt
done; done
for pf in 0 1; do MQ_LN_PREFETCH=$pf timeout 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('vit_b32_image ln_prefetch=$pf  %9.1f emb/s %8.3f ms/step  gemm frac %.3f' % (d['value'], d['ms_per_step'], r['frac']))" | tee -a $out/box_state.txt; done
