#!/bin/bash
# What the codec costs the bench at today's kernels, one box: LM only (--no-mimi: not the metric) / the Mimi chunk serial on the frame
# graphs' stream (--no-overlap-mimi) / beside the next chunk's frame graphs on 128 CUs (the default).  usage: bash tools/ab_codec_share.sh
LOG=gpurun_out/ab_codec_share.log; : > $LOG
B="timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-kernel-timing --no-latency --steps 20 --warmup 5 $*"
for rep in 1 2; do
for opt in "--no-mimi" "--no-overlap-mimi" ""; do
  echo "== rep $rep: ${opt:-default}" >> $LOG
  $B $opt 2>&1 | grep -E "timed|rror|ailed:" >> $LOG || exit 1
done
done
cat $LOG
