# A/B of two builds of the library inside the bench command, one box, one run:  bash tools/ab_lib.sh <other libsmoltts_hip.so> [bench args]
OTHER=$1; shift
run() { timeout -k 10 200 python bench.py --cpu-frames 0 --no-latency "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['us_per_frame_step'], d['roofline']['avg_us'])"; }
for rep in 1 2; do
  echo "== product library"; run "$@"
  echo "== $OTHER"; SMOLTTS_LIB=$OTHER run "$@"
done
