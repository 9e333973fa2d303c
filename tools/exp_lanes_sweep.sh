for cfg in "--batch 2" "--batch 4" "--batch 16" "--cfg" "--res 512"; do for v in off on; do
if [ $v = on ]; then export MKD_DEC_LANES=1; else unset MKD_DEC_LANES; fi
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode 0 --graph 0 $cfg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg lanes $v', round(d['value'],3), round(d['loop']['ms_per_eval'],4))" >> gpurun_out/ab.log || exit 1
done; done
