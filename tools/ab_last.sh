# The fused last SEANet stage (seanet_last.hip) with phase A's MFMAs / the per-half phases compiled out (timing only: wrong
# numerics).  Every arm is a build VARIANT under smoltts_amd/csrc/variants/ loaded through SMOLTTS_LIB; the product library
# is never touched.  Run on the GPU box from the repo root: bash tools/ab_last.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  rm -rf gpurun_out/prof_mimi
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_mimi -o m -- python3 tools/time_mimi.py > gpurun_out/prof_mimi.log 2>&1 < /dev/null
  python3 tools/summarize_kernel_trace.py gpurun_out/prof_mimi | grep seanet_last | cut -c1-150
}
echo "== product"; run
i=0
arms=("-DSMOLTTS_DBG_LAST_NO_A" "-DSMOLTTS_DBG_LAST_NO_HALVES" "-DSMOLTTS_DBG_LAST_NO_ELU" "-DSMOLTTS_DBG_LAST_NO_B" "-DSMOLTTS_DBG_LAST_NO_B -DSMOLTTS_DBG_LAST_NO_ELU")
for flags in "${arms[@]}"; do
  i=$((i+1))
  echo "== flags: $flags"
  export SMOLTTS_LIB=$(python -m smoltts_amd.build --variant ab_last_$i --flags="$flags" | tail -1)
  run
done
