# A/B: how much of a 32-row GEMM launch is the X3 operand?  Builds gemm3 with 3 (product), 2 and 1 activation
# pieces (fewer pieces = wrong numerics, timing only) and runs the LM-only bench + the marginal-cost table.
set -e
for n in 3 2 1; do
  echo "== pieces: $n"
  SMOLTTS_HIPCC_FLAGS="-DSMOLTTS_DBG_PIECES=$n" python -m smoltts_amd.build --force > /dev/null
  timeout -k 10 200 python bench.py --cpu-frames 0 --no-latency --no-mimi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['us_per_frame_step'], d['roofline']['avg_us'])"
done
SMOLTTS_HIPCC_FLAGS="" python -m smoltts_amd.build --force > /dev/null
