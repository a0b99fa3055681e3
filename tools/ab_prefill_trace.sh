cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "" smoltts_amd/csrc/variants/prev3/libsmoltts_hip.so; do
  rm -rf /tmp/pp; 
  if [ -n "$lib" ]; then export SMOLTTS_LIB=$lib; else unset SMOLTTS_LIB; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o p -- python3 tools/profile_prefill.py > /tmp/pp.log 2>&1 < /dev/null
  echo "== lib: ${lib:-product}"; python3 tools/summarize_kernel_trace.py /tmp/pp | grep -E "rows_kernel|attn_prefill|total" | cut -c1-170 | head -8
done
