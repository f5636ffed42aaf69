#!/bin/bash
# round 2, GPU call B: full GPU suite (new: CLIPA filters, stella, Unicode / SentencePiece device tokenisers, bf16 residual stream) +
# GEMM A/Bs (residual epilogue fp32 vs bf16, tile-height sweep) + whole-step A/B of the bf16 residual stream + default bench + ingest profile
TAG=${1:-r02b}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|error|Error|rc=|1-cos|stella|NewModel|residual" $OUT/pytest_gpu.log | tail -30
python tools/gemm_bench.py --only "b32 (out|fc2)" --iters 40 > $OUT/gemm_residual.txt 2>&1; cat $OUT/gemm_residual.txt
python tools/gemm_bench.py --only "b32 (qkv|out|fc1|fc2)$" --iters 30 --rounds 5 --ab "auto:gemm_mt=0;mt2:gemm_mt=2;mt4:gemm_mt=4;mt5:gemm_mt=5;mt6:gemm_mt=6" > $OUT/gemm_mt_sweep.txt 2>&1; cat $OUT/gemm_mt_sweep.txt
for i in 1 2; do
  for v in 0 1; do
    MQ_RESIDUAL_BF16=$v python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('residual_bf16=$v  %9.1f emb/s %8.4f ms/step gemm %6.1f TF frac %.4f  fam %s' % (d['value'], d['ms_per_step'], r['achieved'], r['frac'], {k: round(x['ms_per_step'], 3) for k, x in r['per_family'].items()}))" >> $OUT/residual_ab.txt
  done
done
cat $OUT/residual_ab.txt
for wl in vit_l14_image clip_text_b32; do for v in 0 1; do
  MQ_RESIDUAL_BF16=$v python bench.py --workload $wl --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$wl residual_bf16=$v  %9.1f emb/s %8.4f ms/step gemm %6.1f TF frac %.4f' % (d['value'], d['ms_per_step'], r['achieved'], r['frac']))" >> $OUT/residual_ab.txt
done; done
tail -4 $OUT/residual_ab.txt
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; python -c "
import json; d = json.load(open('$OUT/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac']); print(json.dumps(d.get('e2e_vectorise'), indent=1))"
python - > $OUT/ingest_profile.txt 2>&1 <<'PY'
import cProfile, pstats, sys, os, io
sys.argv = ["bench.py", "--workload", "add_documents_mixed", "--steps", "5", "--warmup", "2"]
import runpy
pr = cProfile.Profile(); pr.enable()
try:
    runpy.run_path("bench.py", run_name="__main__")
finally:
    pr.disable(); s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue())
PY
head -75 $OUT/ingest_profile.txt | tail -60
python tools/latency_bench.py > $OUT/latency.txt 2>&1; tail -15 $OUT/latency.txt
