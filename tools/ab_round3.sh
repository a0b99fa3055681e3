# Round-3 A/B of the frame's launch structure, one box, one run (bench.py, LM + Mimi, no CPU sample):
#   depth layer-0 q|k|v from the engine's table (SMOLTTS_QKV_TABLE=0: wqkv GEMM launches), slow token / last code picked in the
#   commit kernel (SMOLTTS_COMMIT_PICKS=0: launches of their own).  Prints frames/s, us per frame-step, in-situ w1|w3 us.
set -o pipefail
mkdir -p gpurun_out
run() { timeout -k 10 200 python bench.py --cpu-frames 0 --no-latency 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['us_per_frame_step'], d['roofline']['avg_us'])"; }
for rep in 1 2; do
  echo "== table on, picks in commit (default)"; run
  echo "== table off, picks in commit"; SMOLTTS_QKV_TABLE=0 run
  echo "== table on, picks as launches"; SMOLTTS_COMMIT_PICKS=0 run
  echo "== table off, picks as launches (round 2's launch sequence)"; SMOLTTS_QKV_TABLE=0 SMOLTTS_COMMIT_PICKS=0 run
done
