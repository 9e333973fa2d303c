# decoder / encoder lane configurations on one box (B=8, 256x256)
for cfg in "0 0" "2 0" "4 0" "2 1" "4 1" "0 1" "0 0" "2 0" "4 0" "2 1" "4 1"; do set -- $cfg
MKD_DEC_LANES=$1 MKD_ENC_LANES=$2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 --graph ${GRAPH:-0} $EXTRA 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dec_lanes=$1 enc_lanes=$2', round(d['value'],3), round(d['loop']['ms_per_eval'],4))" >> gpurun_out/ab.log || exit 1
done
