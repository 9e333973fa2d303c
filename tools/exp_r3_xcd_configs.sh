# round 3: the XCD-aware order's M <= r N rule (default r = 2) at the other BASELINE configurations: r = 0 (off) / 1 / 2 / 4
mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA 2>gpurun_out/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$* $EXTRA', round(d['value'],3), 'ms/eval', round(d['loop']['ms_per_eval'],3))" || tail -5 gpurun_out/err.log; }
for EXTRA in "--cfg" "--res 512" "--batch 4 --interp 11" "--batch 1" "--batch 16"; do
  for i in 1 2; do
    run MKD_XCD_AUTO_RATIO=0
    run MKD_XCD_AUTO_RATIO=1
    run MKD_XCD_AUTO_RATIO=2
    run MKD_XCD_AUTO_RATIO=4
  done
done
