run() { env "$@" python bench.py --no-cpu-baseline --live-pmc 0 --steps 3 --warmup 1 --decode 0 $EXTRA 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$EXTRA | $* |', round(d['value'],3), 'img/s', round(d['loop']['ms_per_eval'],4), 'ms/eval')"; }
# A/B of environment knobs in the loop, alternating on one box: VARS="MKD_TFM_WARMERS MKD_ATTN_DMA" EXTRA="--res 512" ROUNDS=2
for v in $VARS; do for i in $(seq 1 ${ROUNDS:-2}); do run $v=0; run $v=1; done; done
