# Mimi many-row GEMM on the SEANet shapes with the stores / the MFMAs compiled out (timing only: wrong numerics).
# Every arm is a build VARIANT under smoltts_amd/csrc/variants/ loaded through SMOLTTS_LIB; the product library is never touched.
set -e
timeout -k 10 100 python tools/microbench_rows.py 2>&1 | grep "us "
i=0
for flags in "-DSMOLTTS_DBG_NO_STORE" "-DSMOLTTS_DBG_NO_MFMA" "-DSMOLTTS_DBG_NO_MFMA -DSMOLTTS_DBG_NO_STORE"; do
  i=$((i+1))
  echo "== flags: $flags"
  lib=$(python -m smoltts_amd.build --variant ab_rows_$i --flags="$flags" | tail -1)
  SMOLTTS_LIB=$lib timeout -k 10 100 python tools/microbench_rows.py 2>&1 | grep "us "
done
