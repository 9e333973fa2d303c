# A/B of environment settings against the default on one box, alternating: VARIANTS="A=1 B=2|C=3" (| separates the legs)
mkdir -p gpurun_out
IFS='|' read -ra LEGS <<< "$VARIANTS"
for i in 1 2; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', round(d['value'],3), round(d['loop']['ms_per_eval'],3), d['loop']['launches_per_eval'])"
  for leg in "${LEGS[@]}"; do
    env $leg python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$leg', round(d['value'],3), round(d['loop']['ms_per_eval'],3), d['loop']['launches_per_eval'])"
  done
done
