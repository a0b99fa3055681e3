# A/B of the key-split slow attention inside the bench command at the driver's settings (contexts 325..965), one box, one run
run() { timeout -k 10 300 python bench.py --cpu-frames 0 --no-latency --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['us_per_frame_step'], d['roofline']['avg_us'])"; }
for rep in 1 2; do
  echo "== fp32 KV, split on (default)"; run
  echo "== fp32 KV, split off"; SMOLTTS_SPLIT_ATTN=0 run
done
echo "== bf16 KV, split on"; run --kv bf16
echo "== bf16 KV, split off"; SMOLTTS_SPLIT_ATTN=0 run --kv bf16
