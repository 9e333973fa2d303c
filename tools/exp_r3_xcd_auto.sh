# round 3: XCD-aware tile order only for the weight-heavy GEMMs: MKD_XCD_AUTO_RATIO = r: mode 1 when M <= r N (0 = launch order everywhere).
# (profiles/exp_r3_xcd_auto_m.txt came from the first form of the switch, a plain M threshold MKD_XCD_AUTO_M, which the ratio rule replaced.)
mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA 2>gpurun_out/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$* $EXTRA', round(d['value'],3), 'ms/eval', round(d['loop']['ms_per_eval'],3))" || tail -5 gpurun_out/err.log; }
for i in 1 2; do
  run MKD_XCD_AUTO_RATIO=0
  run MKD_XCD_AUTO_RATIO=1
  run MKD_XCD_AUTO_RATIO=2
  run MKD_XCD_AUTO_RATIO=4
  run MKD_XCD_AUTO_RATIO=8
  run MKD_XCD_AUTO_RATIO=16
done
