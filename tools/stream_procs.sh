#!/bin/bash
# N independent ingest processes (tools/stream_quick.py, pipelined form) on the ONE GPU of the box, started together: does the stream scale with host
# processes (the host side — tokeniser, Pillow export, packing — is what bounds one process)?  usage: bash tools/stream_procs.sh "1 2 3"
for n in ${1:-1 2 3}; do
  for i in $(seq 1 $n); do
    ( timeout 200 python tools/stream_quick.py 2>/dev/null | grep "pipeline_depth=1" | tail -1 | sed "s/^/procs=$n #$i: /" ) &
  done
  wait
done
