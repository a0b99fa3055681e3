# Which weights of a decode frame should carry the non-temporal hint?  Masks of SMOLTTS_STREAM_W_* (include/smoltts_hip.h)
# inside the bench command at the driver's settings, one box, one run.
run() { timeout -k 10 250 python bench.py --cpu-frames 0 --no-latency --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['us_per_frame_step'], d['roofline']['avg_us'])"; }
for m in ${MASKS:-3 0 7 15 23 39 63 3}; do echo "== SMOLTTS_STREAM_W=$m"; SMOLTTS_STREAM_W=$m run; done
