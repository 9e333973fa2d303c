for i in 1 2; do for f in 0 3 6 9; do
MKD_DEC_LANES_FROM=$f python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 --graph 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lanes_from=$f', round(d['value'],3), round(d['loop']['ms_per_eval'],4))" >> gpurun_out/ab.log || exit 1
done; done
