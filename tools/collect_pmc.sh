#!/bin/bash
# HBM traffic of the hot kernels, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 --pmc passes (they do not fit one pass), kernel trace only, the program itself after `--`.
# The program is the bench command itself (no CPU sample, no latency probe, fewer steps: counters serialise every kernel).
# SMOLTTS_MAX_FRAMES_IN_FLIGHT=2: with --pmc every dispatch becomes several packets and the profiler's intercept queue
# does not survive thousands of queued dispatches (round 1 hang) -- the library's bounded run-ahead keeps the host at most
# 2 frames (~460 dispatches) ahead of the GPU; nothing else about the run changes.
# Run on the GPU box from the repo root:  bash tools/collect_pmc.sh [tag]   -> gpurun_out/pmc_{fetch,write}/, pmc_summary.*
set -e
# (raw traces and counter tables stay under /tmp on the GPU box: only summaries go to gpurun_out/, which is merged back up to 64 MiB)
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SMOLTTS_MAX_FRAMES_IN_FLIGHT=2
export SMOLTTS_FRAMES_PER_GRAPH=1   # one frame per graph launch: the bound above counts frames, and a multi-frame graph is that many at once
ARGS="bench.py --cpu-frames 0 --no-latency --no-overlap-mimi --steps 2 --warmup 1"   # (counters serialise every kernel: one stream)
rm -rf /tmp/pmc_fetch /tmp/pmc_write
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -o pmc -- python3 $ARGS > gpurun_out/pmc_fetch.log 2>&1 < /dev/null
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -o pmc -- python3 $ARGS > gpurun_out/pmc_write.log 2>&1 < /dev/null
python3 tools/summarize_pmc.py /tmp/pmc_fetch /tmp/pmc_write gpurun_out/pmc_summary "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of \`python3 $ARGS\` with SMOLTTS_MAX_FRAMES_IN_FLIGHT=2; read = 2 x FETCH_SIZE (gfx950)"
