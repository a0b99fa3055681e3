# A/B sweep of HIP runtime switches around the bench command, one box, one call (the frame is ~180 dependent graph nodes, so what
# the runtime puts between two nodes is first-order):  bash tools/ab_runtime_env.sh [bench args]  > gpurun_out/ab_runtime_env.txt
run() { timeout -k 10 250 python bench.py --cpu-frames 0 --no-latency "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['us_per_frame_step'], d['roofline']['avg_us'])"; }
echo "# frames/s, us per frame-step, in-situ us per w1|w3 launch"
echo "== default"; run "$@" || exit 1
for KV in DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 AMD_OPT_FLUSH=0 HIP_FORCE_DEV_KERNARG=0 HIP_FORCE_DEV_KERNARG=1 \
          ROC_SYSTEM_SCOPE_SIGNAL=0 DEBUG_HIP_KERNARG_COPY_OPT=0 ROC_USE_FGS_KERNARG=0; do
  echo "== $KV"; export $KV; run "$@" || echo "(failed)"; unset ${KV%%=*}
done
echo "== default"; run "$@"
