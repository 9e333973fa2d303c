# A/B an opt-in environment switch on one box: $1 = VAR (set to 1 for the "on" leg), alternating; EXTRA = extra bench flags
for i in 1 2 3; do for v in off on; do
if [ $v = on ]; then export $1=1; else unset $1; fi
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $v', round(d['value'],3), round(d['loop']['ms_per_eval'],4))" >> gpurun_out/ab.log || exit 1
done; done
