# round 3: row split of the single-launch GroupNorm (MKD_GN_HSPLIT = most workgroups per (sample, group chunk); 1 = off)
mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA 2>gpurun_out/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$* $EXTRA', round(d['value'],3), 'ms/eval', round(d['loop']['ms_per_eval'],3), 'groupnorm class ms (events / back to back)', d['kernel_classes_ms_per_eval'].get('groupnorm'), d['kernel_classes_ms_per_eval_back_to_back'].get('groupnorm'))" || tail -5 gpurun_out/err.log; }
for i in 1 2 3; do
  run MKD_GN_HSPLIT=1
  run MKD_GN_HSPLIT=2
  run MKD_GN_HSPLIT=4
  run MKD_GN_HSPLIT=8
done
