export TMPDIR=/tmp
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pf=d['roofline']['per_family']
print('$1', d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(round(v['ms_per_step'],3), v['launches_per_step']) for k,v in pf.items() if k in ('gemm','layernorm','attention')})"; }
timeout 900 python -m pytest tests/test_ln_fold_gpu.py tests/test_towers_gpu.py tests/test_kernels_gpu.py tests/test_gemm_variants_gpu.py -m gpu -q 2>&1 | grep -E "^E  |FAILED|passed|failed" | cut -c1-250 | tail -8
for rep in 1 2; do
MQ_LN_FOLD=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | line "fold=1 (row-stats kernel + epilogue apply)"
MQ_LN_FOLD=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | line "fold=0"
done
for wl in vit_l14_image clip_text_b32 vit_l14_mixed; do for f in 1 0; do
MQ_LN_FOLD=$f timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | line "$wl fold=$f"
done; done
