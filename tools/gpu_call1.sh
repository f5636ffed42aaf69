#!/bin/bash
# GPU parity suite + A/B of the pooled-rows-only last block and the multi-stream knob + kernel trace + PMC traffic passes.
TAG=${1:-c1}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $OUT/pytest_gpu.log
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-28s %9.1f emb/s %8.3f ms/step  e2e %6.1f TF  gemm %6.1f TF (frac %.3f)  fam %s' % ('$1', d['value'], d['ms_per_step'], d['e2e_tflops'], r['achieved'], r['frac'], {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items()}))
"; }
for rs in 0 1; do
  MQ_ROW_SELECT=$rs timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>$OUT/err.txt | line "b32 row_select=$rs" >> $OUT/ab.log 2>&1
done
MARQO_AMD_STREAMS=2 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>$OUT/err.txt | line "b32 streams=2" >> $OUT/ab.log 2>&1
for wl in clip_text_b32 vit_l14_image; do
  for rs in 0 1; do
    MQ_ROW_SELECT=$rs timeout 300 python bench.py --workload $wl --steps 15 --warmup 3 --no-cpu-baseline 2>$OUT/err.txt | line "$wl row_select=$rs" >> $OUT/ab.log 2>&1
  done
done
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o vit_b32 -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/prof.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
done
cd $REPO
python tools/rocpd_summary.py $(ls $OUT/prof/*results.db | head -1) $OUT/kernel_stats.csv 2>>$OUT/prof.err
python tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/traffic.txt 2>&1
rm -rf $OUT/prof/*/*.db.tmp 2>/dev/null
cat $OUT/pytest_gpu.log; cat $OUT/ab.log; cat $OUT/bench.json; head -8 $OUT/kernel_stats.csv | cut -c1-200; cat $OUT/traffic.txt | cut -c1-200
