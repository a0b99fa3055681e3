set -e
for flags in "" "-DSMOLTTS_DBG_NO_STORE" "-DSMOLTTS_DBG_NO_MFMA" "-DSMOLTTS_DBG_NO_MFMA -DSMOLTTS_DBG_NO_STORE"; do
  echo "== flags: $flags"
  SMOLTTS_HIPCC_FLAGS="$flags" python -m smoltts_amd.build --force > /dev/null
  timeout -k 10 100 python tools/microbench_rows.py 2>&1 | grep "us "
done
