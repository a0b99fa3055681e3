set -e
B="timeout -k 10 250 python bench.py --cpu-frames 0 --no-latency"
for opt in "" "--weights fp8" "--batch 64" "--batch 64 --weights fp8" "--batch 128 --weights fp8" "--batch 64 --streams 2 --weights fp8"; do
  echo "== $opt" >> gpurun_out/fp8.log
  $B $opt 2>&1 | grep -E "timed|in-situ|rror|failed" >> gpurun_out/fp8.log
done
cat gpurun_out/fp8.log
