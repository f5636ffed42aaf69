#!/bin/bash
# ONE parametrised GPU-box script (replaces the per-call gpu_r0x_*.sh of rounds 2-3):  gpurun -- 'bash tools/gpu_run.sh <tag> <stage> [<stage> ...]'
# Results land in gpurun_out/<tag>/; the summaries that are cited get copied to profiles/ by hand.  Stages:
#   tests[:<pytest -k expr>]      pytest -m gpu (whole suite, or the selection)
#   bench[:<workload>]            one bench.py line (default workload, with extras) -> bench_<workload>.json
#   ab:<VAR>=<a>,<b>[:<workload>] headline (or <workload>) twice per value of an environment knob, interleaved, --no-extras
#   prof                          rocprofv3 --kernel-trace --stats of the headline command -> kernel_stats.csv
#   pmc                           FETCH_SIZE / WRITE_SIZE / SQ passes (separate runs, as MI355X_MICROARCH.md prescribes) -> traffic.txt, sq.txt
#   e2e                           tools/e2e_quick.py (one synchronous 256-image PIL caller)
#   py:<script and args>          python <script and args>
TAG=${1:-run}; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
NPY=0
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pf=d['roofline']['per_family']
print('$1', d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(round(v['ms_per_step'],3), v['launches_per_step']) for k,v in pf.items() if k in ('gemm','layernorm','attention')})"; }
for stage in "$@"; do
  kind=${stage%%:*}; arg=${stage#*:}; [ "$arg" = "$stage" ] && arg=""
  case $kind in
    tests)
      if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -q -x -s -k "$arg" 2>&1 | grep -v "^$" | tail -25 | cut -c1-800 | tee $OUT/pytest_sel.txt
      else timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt; fi ;;
    bench)
      wl=${arg:-vit_b32_image}
      timeout 900 python bench.py --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; tail -c 1500 $OUT/bench_$wl.json; tail -3 $OUT/bench_$wl.err ;;
    ab)
      spec=${arg%%:*}; wl=${arg#*:}; [ "$wl" = "$arg" ] && wl=vit_b32_image
      var=${spec%%=*}; vals=${spec#*=}
      for rep in 1 2; do for v in ${vals//,/ }; do
        env $var=$v timeout 300 python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>$OUT/ab.err | line "$wl $var=$v" | tee -a $OUT/ab_$var.txt
      done; done ;;
    prof)
      cd /tmp
      timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o vit_b32 -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_prof.json 2> $OUT/prof.err
      cd $REPO
      python tools/rocpd_summary.py $(ls $OUT/prof/*results.db | head -1) $OUT/kernel_stats.csv 2>>$OUT/prof.err
      rm -rf $OUT/prof; head -12 $OUT/kernel_stats.csv | cut -c1-180 ;;
    pmc)
      cd /tmp
      for c in FETCH_SIZE WRITE_SIZE; do
        timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
      done
      timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_SQ -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_SQ.json 2> $OUT/pmc_SQ.err
      cd $REPO
      python tools/pmc_sq_summary.py $OUT/pmc_SQ > $OUT/sq.txt 2>&1
      python tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE --json $OUT/traffic.json > $OUT/traffic.txt 2>&1
      rm -f $OUT/pmc_FETCH_SIZE/*.db $OUT/pmc_WRITE_SIZE/*.db $OUT/pmc_SQ/*.db
      grep -E "gemm|layernorm|attention|patchify" $OUT/traffic.txt | cut -c1-140; cat $OUT/sq.txt ;;
    e2e) timeout 300 python tools/e2e_quick.py 2>/dev/null | tail -1 | tee -a $OUT/e2e.txt ;;
    py) NPY=$((NPY+1)); timeout 900 python $arg 2>&1 | tail -60 | tee $OUT/py_$NPY.txt ;;
    *) echo "unknown stage $stage" ;;
  esac
done
