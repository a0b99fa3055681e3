set -e
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do timeout -k 10 200 python3 bench.py --cpu-frames 0 --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['us_per_frame_step'], d['roofline']['avg_us'], d['first_audio_chunk']['b1_70m_stream_frames_per_s'], d['first_audio_chunk']['b1_70m_stream_frames_per_s_one_stream'], d['first_audio_chunk']['b1_70m_ms_p50'])"; done
