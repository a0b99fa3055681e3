# A/B of an environment switch inside the bench command, one box, one run:  bash tools/ab_env.sh VAR=VALUE [bench args]
# (e.g. SMOLTTS_STREAM_SLOW_W=0, SMOLTTS_SPLIT_ATTN=0, SMOLTTS_QKV_TABLE=0, SMOLTTS_COMMIT_PICKS=0, HIP_FORCE_DEV_KERNARG=1)
KV=$1; shift
run() { timeout -k 10 250 python bench.py --cpu-frames 0 --no-latency "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['us_per_frame_step'], d['roofline']['avg_us'])"; }
for rep in 1 2; do
  echo "== default"; run "$@"
  echo "== $KV"; env $KV bash -c "$(declare -f run); run $*"
done
