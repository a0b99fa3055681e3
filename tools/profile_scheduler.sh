#!/bin/bash
# Which kernels stretch when the scheduler runs codec passes beside the ticks?  rocprofv3 --kernel-trace of
# tools/bench_scheduler.py (streamed responses), summarised per launch shape; compare the averages with the same shapes in
# profiles/<tag>_bench_kernel_breakdown.txt (the fixed-batch bench, no concurrent codec).  The raw trace stays in /tmp.
# Run on the GPU box from the repo root:  bash tools/profile_scheduler.sh [n_requests] [tick]
set -e
N=${1:-160}; TICK=${2:-4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_sched
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_sched -o sched -- python3 tools/bench_scheduler.py $N $TICK stream > gpurun_out/prof_sched.log 2>&1 < /dev/null
python3 tools/summarize_kernel_trace.py /tmp/prof_sched > gpurun_out/prof_sched_breakdown.txt
head -16 gpurun_out/prof_sched_breakdown.txt
