#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command (CPU sample and latency probe off; the Mimi decode on the frame graphs' stream:
# per-kernel durations undisturbed, and rocprofv3 + a CU-masked stream (the default's hipExtStreamCreateWithCUMask) segfaults at exit) -> gpurun_out/prof_<tag>/,
# plus the per-launch-shape breakdown.  Copy the two summaries into profiles/ (tools/README.md).
# Run on the GPU box from the repo root:  bash tools/collect_profile.sh [tag]
set -e
# (raw traces and counter tables stay under /tmp on the GPU box: only summaries go to gpurun_out/, which is merged back up to 64 MiB)
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_$TAG
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python3 bench.py --cpu-frames 0 --no-latency --no-overlap-mimi > gpurun_out/prof_$TAG.log 2>&1 < /dev/null
python3 tools/summarize_kernel_trace.py /tmp/prof_$TAG > gpurun_out/prof_${TAG}_breakdown.txt
find /tmp/prof_$TAG -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_${TAG}_kernel_stats.csv \;
head -30 gpurun_out/prof_${TAG}_breakdown.txt
