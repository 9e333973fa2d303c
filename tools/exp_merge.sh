for i in 1 2 3; do for v in 0 1; do
MKD_MERGE_FFOUT=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 --graph 0 $EXTRA 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('merge_ffout=$v', round(d['value'],3), round(d['loop']['ms_per_eval'],4), d['loop']['launches_per_eval'])" >> gpurun_out/ab.log || exit 1
done; done
