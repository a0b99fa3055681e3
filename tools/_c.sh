run() { echo "== $*"; timeout -k 10 250 python3 bench.py --cpu-frames 0 --no-latency --no-kernel-timing --steps 4 --warmup 1 "$@" 2>gpurun_out/c.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['codec'][:60], d['parity']['ranks_passed'])" || { echo FAILED; tail -5 gpurun_out/c.err; }; }
run --batch 64
run --weights fp8 --batch 64 --prefill-chunk 128
run --kv bf16
run --streams 2
run --no-mimi
run --batch 8
run --model smoltts_byte_70m --batch 1
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_lm_gpu.py tests/test_attn_split_gpu.py -x -q 2>&1 | tail -2
