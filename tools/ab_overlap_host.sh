# A/B: Mimi decode on a second stream beside the frame graphs: serial order, device-side wait_event, and the hand-over done
# by the host (no blocked barrier packet in the second queue).  Round 1: 21.9k / 19.4k / 22.3k frames/s.
set -e
run() { timeout -k 10 200 python bench.py --cpu-frames 0 --no-latency --no-kernel-timing "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['us_per_frame_step'])"; }
echo "== serial"; run
echo "== overlap, device wait"; run --overlap-mimi
echo "== overlap, host wait"; run --overlap-mimi --overlap-wait host
echo "== overlap, host wait, Mimi on 64 CUs (xcd), LM on the others"; run --overlap-mimi --overlap-wait host --mimi-cus 64 --cu-pattern xcd --lm-complement
