for i in 1 2; do for g in 0 1; do
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --graph $g --decode 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph=$g decode=1', round(d['value'],3), round(d['ms_per_step'],2))" >> gpurun_out/ab.log || exit 1
done; done
