mkdir -p gpurun_out/r08d
for cfg in "1 0" "1 100" "2 0" "2 100" "2 300" "3 100" "4 100" "4 0"; do
  set -- $cfg
  echo "== depth=$1 window=$2" >> gpurun_out/r08d/sweep.txt
  MARQO_AMD_NATIVE_QUEUE_DEPTH=$1 MARQO_AMD_NATIVE_QUEUE_WINDOW_US=$2 timeout 200 python tools/queue_bench.py --only ViT-B-32 --calls 60 2>/dev/null | grep "raw\|vectorise" >> gpurun_out/r08d/sweep.txt
done
cat gpurun_out/r08d/sweep.txt
