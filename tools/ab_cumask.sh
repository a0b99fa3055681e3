set -e
B="timeout -k 10 200 python bench.py --cpu-frames 0 --no-kernel-timing --no-latency"
for opt in "" "--overlap-mimi" "--overlap-mimi --mimi-cus 64" "--overlap-mimi --mimi-cus 64 --lm-complement" "--overlap-mimi --mimi-cus 64 --cu-pattern xcd --lm-complement" "--overlap-mimi --mimi-cus 32 --lm-complement" "--overlap-mimi --mimi-cus 96 --lm-complement"; do
  echo "== $opt" >> gpurun_out/cumask.log
  $B $opt 2>&1 | grep -E "timed|CU masks|rror|failed" >> gpurun_out/cumask.log
done
cat gpurun_out/cumask.log
