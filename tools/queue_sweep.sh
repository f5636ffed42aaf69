# the native queue's shape under light and heavy load: depth x helper_seqs x window (tools/queue_bench.py, CLIP text B/32, vectorise_ndarray() rows)
TAG=$1; mkdir -p gpurun_out/$TAG
for cfg in "1 0 0" "2 4 0" "2 8 0" "3 4 0" "1 0 0" "2 4 0"; do
  set -- $cfg
  for th in 2 4 8 16; do
    echo "== depth=$1 helper_seqs=$2 window=$3 threads=$th" >> gpurun_out/$TAG/sweep.txt
    MARQO_AMD_NATIVE_QUEUE_DEPTH=$1 MARQO_AMD_NATIVE_QUEUE_HELPER_SEQS=$2 MARQO_AMD_NATIVE_QUEUE_WINDOW_US=$3 timeout 200 python tools/queue_bench.py --only ViT-B-32 --threads $th --calls 80 2>/dev/null | grep "vectorise" | sed 's/open_clip.ViT-B-32.laion2b_s34b_b79k //; s/; coalescer.*//' >> gpurun_out/$TAG/sweep.txt
  done
done
cat gpurun_out/$TAG/sweep.txt
