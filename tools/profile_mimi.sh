#!/bin/bash
# Per-kernel times of the Mimi chunk decode (tools/time_mimi.py) under rocprofv3 --kernel-trace.  Run on the GPU box from the repo root.
set -e
# (raw traces and counter tables stay under /tmp on the GPU box: only summaries go to gpurun_out/, which is merged back up to 64 MiB)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_mimi
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_mimi -o m -- python3 tools/time_mimi.py "$@" > gpurun_out/prof_mimi.log 2>&1 < /dev/null
python3 tools/summarize_kernel_trace.py /tmp/prof_mimi | cut -c1-175 | head -${LINES_OUT:-32}
