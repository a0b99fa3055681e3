set -e
B="timeout -k 10 250 python bench.py --cpu-frames 0 --no-latency --no-kernel-timing"
for opt in "--batch 64" "--batch 96" "--batch 128" "--batch 128 --weights fp8"; do
  echo "== $opt" >> gpurun_out/b64.log
  $B $opt 2>&1 | grep -E "timed|rror|failed" >> gpurun_out/b64.log
done
cat gpurun_out/b64.log
