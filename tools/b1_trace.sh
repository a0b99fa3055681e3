# Kernel trace of the B = 1 smoltts_byte_70m stream (tools/profile_b1.py): us per frame by launch class.  bash tools/b1_trace.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/profile_b1.py 128 > gpurun_out/r4_b1_walltimes.txt 2>&1
cat gpurun_out/r4_b1_walltimes.txt
rm -rf /tmp/b1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/b1 -o p -- python3 tools/profile_b1.py 64 > /tmp/b1.log 2>&1 < /dev/null
python3 tools/summarize_kernel_trace.py /tmp/b1 | cut -c1-170 | head -60 > gpurun_out/r4_b1_kernel_trace.txt
head -45 gpurun_out/r4_b1_kernel_trace.txt
