# A/B of an environment switch inside the bench command, alternating, one box:  bash tools/ab_env2.sh VAR=VALUE [bench args]
KV=$1; shift
run() { timeout -k 10 250 python bench.py --cpu-frames 0 --no-latency "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['us_per_frame_step'], d['roofline']['avg_us'])"; }
for rep in 1 2; do
  echo "== default"; run "$@"
  echo "== $KV"; export $KV; run "$@"; unset ${KV%%=*}
done
