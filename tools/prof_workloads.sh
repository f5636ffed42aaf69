# rocprofv3 --kernel-trace --stats of bench.py --workload <wl> for each workload named (default: the two text towers) -> gpurun_out/$PROF_TAG/kernel_stats_<wl>.csv (PROF_TAG, default r06k)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/${PROF_TAG:-r06k}; cd /tmp; export TMPDIR=/tmp
for wl in ${@:-bert_base_77 clip_text_b32}; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${PROF_TAG:-r06k}/prof_$wl -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/${PROF_TAG:-r06k}/bench_$wl.json 2> $GRAFT_REPO_ROOT/gpurun_out/${PROF_TAG:-r06k}/prof_$wl.err
  cd $GRAFT_REPO_ROOT; python tools/rocpd_summary.py $(ls gpurun_out/${PROF_TAG:-r06k}/prof_$wl/*results.db | head -1) gpurun_out/${PROF_TAG:-r06k}/kernel_stats_$wl.csv 2>>gpurun_out/${PROF_TAG:-r06k}/prof_$wl.err; rm -rf gpurun_out/${PROF_TAG:-r06k}/prof_$wl; cd /tmp
  head -14 $GRAFT_REPO_ROOT/gpurun_out/${PROF_TAG:-r06k}/kernel_stats_$wl.csv | cut -c1-60,200-330
done
