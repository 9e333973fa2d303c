for ov in 0 1 0 1; do for g in 0 1; do
MKD_DEC_OVERLAP=$ov python bench.py --steps 3 --warmup 1 --no-cpu-baseline --graph $g --decode 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ov=$ov graph=$g', round(d['value'],3), round(d['loop']['ms_per_eval'],4))" >> gpurun_out/ab.log || exit 1
done; done
