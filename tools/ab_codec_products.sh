#!/bin/bash
# The codec's matrix-core kernels on three instead of six bf16x3 products per operand pair (bench.py --codec-products 3 =
# SMOLTTS_MIMI_OPT_PRODUCTS), alternating with the default on one box.  usage: bash tools/ab_codec_products.sh [bench args]
LOG=gpurun_out/ab_codec_products.log; : > $LOG
B="timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-kernel-timing --no-latency --steps 20 --warmup 5 $*"
for rep in 1 2; do
for opt in "" "--codec-products 3"; do
  echo "== rep $rep: ${opt:-default (six products)}" >> $LOG
  $B $opt 2> $LOG.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'frames/s;', d['ms_per_step'], 'ms per step')" >> $LOG || { tail -5 $LOG.err >> $LOG; exit 1; }
done
done
cat $LOG
