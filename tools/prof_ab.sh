#!/bin/bash
# rocprofv3 kernel stats of one bench workload under two values of an environment knob:  bash tools/prof_ab.sh <tag> <workload> <VAR> <a> <b>
TAG=$1; WL=$2; VAR=$3; A=$4; B=$5
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp
for v in $A $B; do
  cd /tmp
  env $VAR=$v timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$v -o p -- python $REPO/bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_$v.json 2> $OUT/prof_$v.err
  cd $REPO
  python tools/rocpd_summary.py $(ls $OUT/prof_$v/*results.db | head -1) $OUT/kernel_stats_$v.csv 2>>$OUT/prof_$v.err
  rm -rf $OUT/prof_$v
  echo "== $VAR=$v"; head -14 $OUT/kernel_stats_$v.csv | cut -c1-75,200-330
done
