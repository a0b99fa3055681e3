#!/bin/bash
# HBM traffic of one Mimi chunk decode (tools/time_mimi.py: 32 slots x 32 frames, 21 chunk decodes per run): FETCH_SIZE and
# WRITE_SIZE in separate rocprofv3 --pmc passes, per kernel, then the per-chunk sum.  Run on the GPU box from the repo root:
#   bash tools/pmc_mimi.sh [tag]   -> gpurun_out/pmc_mimi_<tag>.txt
set -e
# (raw traces and counter tables stay under /tmp on the GPU box: only summaries go to gpurun_out/, which is merged back up to 64 MiB)
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pmcm_fetch /tmp/pmcm_write
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcm_fetch -o pmc -- python3 tools/time_mimi.py > gpurun_out/pmcm_fetch.log 2>&1 < /dev/null
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmcm_write -o pmc -- python3 tools/time_mimi.py > gpurun_out/pmcm_write.log 2>&1 < /dev/null
python3 tools/summarize_pmc.py /tmp/pmcm_fetch /tmp/pmcm_write gpurun_out/pmcm_summary "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/time_mimi.py; read = 2 x FETCH_SIZE (gfx950)" > /dev/null
python3 tools/pmc_mimi_total.py gpurun_out/pmcm_summary.txt 21 > gpurun_out/pmc_mimi_$TAG.txt
cat gpurun_out/pmc_mimi_$TAG.txt
