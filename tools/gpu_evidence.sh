#!/bin/bash
# Evidence run: GPU test suite, headline bench line (with CPU baseline), rocprofv3 kernel trace of the same command, and the
# (the profiled passes run the headline workload alone, --no-extras, so that per-kernel averages and PMC sums are not mixed with the `also` workloads)
# two PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes) for roofline.traffic.
# usage: tools/gpu_evidence.sh <tag> [skiptests]
TAG=${1:-ev}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
if [ "$2" != "skiptests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/pytest_gpu.log
fi
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o vit_b32 -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_prof.json 2> $OUT/prof.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
done
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_SQ -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_SQ.json 2> $OUT/pmc_SQ.err
cd $REPO
python tools/pmc_sq_summary.py $OUT/pmc_SQ > $OUT/sq.txt 2>&1
python tools/rocpd_summary.py $(ls $OUT/prof/*results.db | head -1) $OUT/kernel_stats.csv 2>>$OUT/prof.err
python tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE --json $OUT/traffic.json > $OUT/traffic.txt 2>&1
cp $OUT/traffic.json profiles/r03_traffic_vit_b32_image_bf16.json   # bench.py reads roofline.traffic from here
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
rm -rf $OUT/prof $OUT/pmc_FETCH_SIZE/*.db $OUT/pmc_WRITE_SIZE/*.db $OUT/pmc_SQ/*.db 2>/dev/null
cat $OUT/pytest_gpu.log 2>/dev/null; cat $OUT/bench.json; head -9 $OUT/kernel_stats.csv | cut -c1-180; grep -E "gemm|layernorm|attention|patchify" $OUT/traffic.txt | cut -c1-140; cat $OUT/sq.txt
