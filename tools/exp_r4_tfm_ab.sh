# round 4: the fused transformer tail in the loop: off / forced on / shape policy, alternating on one box, at the BASELINE configurations
mkdir -p gpurun_out
OUT=gpurun_out/exp_r4_tfm_ab.txt; : > $OUT
run() { env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA 2>gpurun_out/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$EXTRA | $* |', round(d['value'],3), 'img/s', round(d['loop']['ms_per_eval'],3), 'ms/eval', d['loop']['launches_per_eval'], 'launches')" >> $OUT || tail -5 gpurun_out/err.log >> $OUT; tail -1 $OUT; }
for EXTRA in "" "--res 512" "--cfg" "--batch 4 --interp 11"; do
  for i in 1 2; do
    run MKD_TFM_TAIL=0
    run MKD_TFM_TAIL=1
    run MKD_TFM_TAIL=-1 ${POLICY_ENV:-MKD_X=0}
  done
done
