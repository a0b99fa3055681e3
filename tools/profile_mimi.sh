#!/bin/bash
# Per-kernel times of the Mimi chunk decode (tools/time_mimi.py) under rocprofv3 --kernel-trace.  Run on the GPU box from the repo root.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_mimi
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_mimi -o m -- python3 tools/time_mimi.py "$@" > gpurun_out/prof_mimi.log 2>&1 < /dev/null
python3 tools/summarize_kernel_trace.py gpurun_out/prof_mimi | cut -c1-175 | head -${LINES_OUT:-32}
