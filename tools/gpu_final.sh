#!/bin/bash
# Round-end evidence run: GPU test suite, headline bench line (with CPU baseline), rocprofv3 kernel trace of the same command,
# and the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs as MI355X_MICROARCH.md prescribes) for roofline.traffic.
TAG=${1:-final}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o vit_b32 -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/prof.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
done
cd $REPO
python tools/rocpd_summary.py $(ls $OUT/prof/*results.db | head -1) $OUT/kernel_stats.csv 2>>$OUT/prof.err
python tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/traffic.txt 2>&1
cat $OUT/pytest_gpu.log; cat $OUT/bench.json; head -8 $OUT/kernel_stats.csv | cut -c1-200; cat $OUT/traffic.txt
