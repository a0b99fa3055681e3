cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o p -- python3 tools/profile_prefill.py > /tmp/pp.log 2>&1 < /dev/null
tail -3 /tmp/pp.log
python3 tools/summarize_kernel_trace.py /tmp/pp | cut -c1-170 | head -24
