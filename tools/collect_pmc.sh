#!/bin/bash
# HBM traffic of the hot kernels, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 --pmc passes (they do not fit one pass), kernel trace only, the program itself after `--`.
# Run on the GPU box from the repo root:  bash tools/collect_pmc.sh   -> gpurun_out/pmc_{fetch,write}/, pmc_summary.*
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# only the decode-loop kernels are instrumented (rocprofv3's counter service crashed on the prefill launches)
FILTER='gemm3_kernel|attn_kernel|attn_short_kernel|argmax_kernel'
ARGS="tools/pmc_decode.py"
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$FILTER" --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o pmc -- python3 $ARGS > gpurun_out/pmc_fetch.log 2>&1 < /dev/null
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$FILTER" --kernel-trace --output-format csv -d gpurun_out/pmc_write -o pmc -- python3 $ARGS > gpurun_out/pmc_write.log 2>&1 < /dev/null
python3 tools/summarize_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_summary
