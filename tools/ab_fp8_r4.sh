# Round-4 numbers of BASELINE configs[4] (fp8 weights + chunked prefill, B = 64) and of fp8 against bf16 at B = 32 / 64, one box, one run.
# Full bench lines (JSON) are kept: gpurun_out/r4_fp8_<name>.json; the summary goes to stdout.   bash tools/ab_fp8_r4.sh
mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 300 python bench.py --cpu-frames 0 --no-latency --steps 20 --warmup 5 "$@" 2>/dev/null > gpurun_out/r4_fp8_$name.json; python -c "import sys,json; d=json.loads(open('gpurun_out/r4_fp8_$name.json').read()); print('$name', d['value'], d['us_per_frame_step'], d['roofline']['avg_us'] if d.get('roofline') else None, d['prefill_ms'])"; }
for rep in 1 2; do
  run bf16_b32
  run fp8_b32 --weights fp8
  run bf16_b64 --batch 64
  run fp8_b64 --weights fp8 --batch 64
  run fp8_b64_chunked --weights fp8 --batch 64 --prefill-chunk 128
done
