# round 3: grouped encoder launches (MKD_ENC_GROUP) and the per-call time-embedding table (MKD_TEMB_TABLE) against the round-2
# structure, alternating on one box; EXTRA = extra bench flags
mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA 2>gpurun_out/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_classes_ms_per_eval']; print('$*', round(d['value'],3), 'ms/eval', round(d['loop']['ms_per_eval'],3), 'launches', d['loop']['launches_per_eval'], 'serial sum', round(sum(k.values()),3))" || tail -5 gpurun_out/err.log; }
for i in 1 2; do
  run MKD_ENC_GROUP=0 MKD_TEMB_TABLE=0
  run MKD_ENC_GROUP=0 MKD_TEMB_TABLE=1
  run MKD_ENC_GROUP=1 MKD_TEMB_TABLE=0
  run MKD_ENC_GROUP=1 MKD_TEMB_TABLE=1
  run MKD_ENC_GROUP=1 MKD_TEMB_TABLE=1 MKD_DEC_LANES=0
  run MKD_ENC_GROUP=1 MKD_TEMB_TABLE=1 MKD_DEC_LANES=0 MKD_DEC_OVERLAP=0
  run MKD_ENC_GROUP=1 MKD_TEMB_TABLE=1 MKD_LANE_HELPERS=0
done
