#!/bin/bash
# round 4: the Mimi chunk decode beside the frame graphs at today's kernels -- serial (default) / second stream with a
# device-side wait / second stream handed over by the host / the same with the Mimi stream on a subset of the CUs.
# usage (GPU box, repo root): bash tools/ab_overlap_r4.sh [bench args]
LOG=gpurun_out/ab_overlap_r4.log; : > $LOG
B="timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-kernel-timing --no-latency --steps 20 --warmup 5 $*"
for rep in 1 2; do
for opt in "" "--overlap-mimi" "--overlap-mimi --overlap-wait host" "--overlap-mimi --overlap-wait host --mimi-cus 128" "--overlap-mimi --overlap-wait host --mimi-cus 64" "--overlap-mimi --overlap-wait host --mimi-cus 128 --cu-pattern xcd"; do
  echo "== rep $rep: $opt" >> $LOG
  $B $opt 2>&1 | grep -E "timed|CU masks|rror|failed" >> $LOG || exit 1
done
done
cat $LOG
