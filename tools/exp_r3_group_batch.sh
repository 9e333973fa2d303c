# round 3: grouped encoder chain vs two chains at other batch sizes / shapes (latency-bound small batches, throughput-bound large ones)
mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA 2>gpurun_out/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$* $EXTRA', round(d['value'],3), 'ms/eval', round(d['loop']['ms_per_eval'],3), 'launches', d['loop']['launches_per_eval'])" || tail -5 gpurun_out/err.log; }
for b in 1 2 4 16; do for i in 1 2; do
  EXTRA="--batch $b" run MKD_ENC_GROUP=0
  EXTRA="--batch $b" run MKD_ENC_GROUP=1
  EXTRA="--batch $b" run MKD_ENC_GROUP=1 MKD_DEC_LANES=0
done; done
for i in 1 2; do
  EXTRA="--cfg" run MKD_ENC_GROUP=0
  EXTRA="--cfg" run MKD_ENC_GROUP=1
  EXTRA="--res 512" run MKD_ENC_GROUP=0
  EXTRA="--res 512" run MKD_ENC_GROUP=1
done
