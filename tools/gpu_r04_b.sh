#!/bin/bash
# round 4, call B: LayerNorm folded into the QKV / fc1 GEMMs on the bf16 stream (row statistics from the staged A tiles) + weight prefetch on the attention launch
OUT=$PWD/gpurun_out/r04b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ln_fold_gpu.py tests/test_towers_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -40 > $OUT/pytest_fold.txt; tail -25 $OUT/pytest_fold.txt
for rep in 1 2; do
for fold in 0 1; do
  MQ_LN_FOLD=$fold timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras > $OUT/bench_fold$fold.json 2>$OUT/bench_fold$fold.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_fold$fold.json").read().strip().splitlines()[-1])
pf=d["roofline"]["per_family"]
print("fold=$fold", d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:(round(v["ms_per_step"],3), v["launches_per_step"]) for k,v in pf.items()})
PY
done
done
for wl in vit_l14_image clip_text_b32; do
for fold in 0 1; do
  MQ_LN_FOLD=$fold timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_${wl}_fold$fold.json 2>$OUT/bench_${wl}_fold$fold.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_${wl}_fold$fold.json").read().strip().splitlines()[-1])
pf=d["roofline"]["per_family"]
print("$wl fold=$fold", d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:(round(v["ms_per_step"],3), v["launches_per_step"]) for k,v in pf.items()})
PY
done
done
MQ_LN_FOLD=1 MQ_LN_PREFETCH=0 timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold=1 prefetch=0', d['value'], d['ms_per_step'])"
MQ_LN_FOLD=0 MQ_LN_PREFETCH=0 timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold=0 prefetch=0', d['value'], d['ms_per_step'])"
