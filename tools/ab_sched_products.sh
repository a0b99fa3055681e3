#!/bin/bash
# Serving (tools/bench_scheduler.py, 512 requests, 32 slots) with the codec on six / three bf16x3 products: streamed at 4- and 8-frame ticks,
# blocking at 4-frame ticks; alternating, one box.  usage: bash tools/ab_sched_products.sh
LOG=gpurun_out/ab_sched_products.log; : > $LOG
for mode in "4 stream" "8 stream" "4 block"; do
for prod in 6 3; do
  echo "== $mode, CODEC_PRODUCTS=$prod" >> $LOG
  CODEC_PRODUCTS=$prod timeout -k 10 200 python3 tools/bench_scheduler.py 512 $mode 2>/dev/null | grep -E "frames/s|GPU time" >> $LOG || exit 1
done
done
cat $LOG
