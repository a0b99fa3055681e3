# Round-4 A/B of the depth attention inside the wo launch (SMOLTTS_OPT_FUSE_DEPTH_ATTN), one box, one run (bench.py, LM + Mimi,
# no CPU sample, no latency probe).  Prints frames/s, us per frame-step, in-situ w1|w3 us.   bash tools/ab_round4.sh [bench args]
set -o pipefail
mkdir -p gpurun_out
run() { timeout -k 10 250 python bench.py --cpu-frames 0 --no-latency "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['us_per_frame_step'], d['roofline']['avg_us'])"; }
for rep in 1 2; do
  echo "== depth attention inside wo (default)"; run "$@"
  echo "== depth attention as launches of its own"; SMOLTTS_FUSE_DEPTH_ATTN=0 run "$@"
done
