# round 4: the fused transformer tail's row threshold at the small batches (default 4096 rows)
mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA 2>gpurun_out/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$EXTRA | $* |', round(d['value'],3), 'img/s', round(d['loop']['ms_per_eval'],3), 'ms/eval', d['loop']['launches_per_eval'])" || tail -5 gpurun_out/err.log; }
for EXTRA in "--batch 4" "--batch 2" "--batch 1" "--batch 8"; do for i in 1 2; do for r in 8192 4096 2048 1024; do run MKD_TFM_TAIL_MINROWS=$r; done; done; done
