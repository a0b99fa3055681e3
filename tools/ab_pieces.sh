# A/B: how much of a 32-row GEMM launch is the X3 operand?  Builds gemm3 with 3 (product), 2 and 1 activation
# pieces (fewer pieces = wrong numerics, timing only) and runs the LM-only bench.  The 2- and 1-piece arms are build
# VARIANTS under smoltts_amd/csrc/variants/ loaded through SMOLTTS_LIB; the product library is never touched.
set -e
run() { timeout -k 10 200 python bench.py --cpu-frames 0 --no-latency --no-mimi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['us_per_frame_step'], d['roofline']['avg_us'])"; }
echo "== pieces: 3 (product)"; run
for n in 2 1; do
  echo "== pieces: $n"
  lib=$(python -m smoltts_amd.build --variant pieces_$n --flags "-DSMOLTTS_DBG_PIECES=$n" | tail -1)
  SMOLTTS_LIB=$lib run
done
# the residual GEMMs alone (wo, w2: VERDICT r02 item 1c -- an upper bound on what a smaller activation operand for them can buy:
# one piece = a third of their X3 bytes AND a third of their MFMAs)
for n in 2 1; do
  echo "== pieces of wo / w2 only: $n"
  lib=$(python -m smoltts_amd.build --variant pieces_resid_$n --flags "-DSMOLTTS_DBG_PIECES_RESID=$n" | tail -1)
  SMOLTTS_LIB=$lib run
done
