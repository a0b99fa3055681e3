# A/B of two builds on the first-audio-chunk latency (prefill + frame 0 + Mimi step + copy), inside the bench command:
#   bash tools/ab_first_chunk.sh <other libsmoltts_hip.so>
OTHER=$1
run() { timeout -k 10 250 python bench.py --cpu-frames 0 --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['first_audio_chunk']; print('prefill_ms', d['prefill_ms'], 'first chunk p50: 32 at once', f['batch_at_once_ms_p50'], 'steady', f['steady_one_arrival_ms_p50'], 'b1 70m', f['b1_70m_ms_p50'], 'value', d['value'])"; }
for rep in 1 2; do
  echo "== product library"; run
  echo "== $OTHER"; SMOLTTS_LIB=$OTHER run
done
