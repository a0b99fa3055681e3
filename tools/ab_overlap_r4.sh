#!/bin/bash
# round 4: the Mimi chunk decode beside the frame graphs at today's kernels -- serial (--no-overlap-mimi) / second stream with a
# device-side wait / second stream handed over by the host / the same with the Mimi stream on 128 CUs (the default) / 64 / whole XCDs.
# usage (GPU box, repo root): bash tools/ab_overlap_r4.sh [bench args]
LOG=gpurun_out/ab_overlap_r4.log; : > $LOG
B="timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-kernel-timing --no-latency --steps 20 --warmup 5 $*"
for rep in 1 2; do
for opt in "--no-overlap-mimi" "--overlap-wait device --mimi-cus 0" "--mimi-cus 0" "" "--mimi-cus 64" "--cu-pattern xcd"; do
  echo "== rep $rep: $opt" >> $LOG
  $B $opt 2>&1 | grep -E "timed|CU masks|rror|ailed:" >> $LOG || exit 1
done
done
cat $LOG
