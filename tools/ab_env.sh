# A/B of HIP runtime environment knobs on the LM-only bench (launch-latency-bound frame graphs).
# Round 1 (MI355X, ROCm 7.2): no effect — 1250 / 1251 / 1254 / 1255 us per LM frame-step for default / DEV_KERNARG=1 / =0 / MAX_HW_QUEUES=2.
set -e
run() { timeout -k 10 200 python bench.py --cpu-frames 0 --no-latency --no-mimi --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['us_per_frame_step'])"; }
echo "== default"; run
echo "== HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 run
echo "== HIP_FORCE_DEV_KERNARG=0"; HIP_FORCE_DEV_KERNARG=0 run
echo "== GPU_MAX_HW_QUEUES=2"; GPU_MAX_HW_QUEUES=2 run
echo "== default"; run
