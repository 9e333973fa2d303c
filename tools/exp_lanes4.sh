# 2 vs 4 decoder lanes on the larger effective batches (CFG = 16 samples / evaluation, batch 16, 512x512, interpolation 44)
mkdir -p gpurun_out
run() { name=$1; shift; l=$1; shift; env MKD_DEC_LANES=$l python bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode 0 "$@" > gpurun_out/l4_$name.json 2> gpurun_out/l4_$name.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/l4_$name.json")); print("$name", round(d["value"],3), round(d["loop"]["ms_per_eval"],3), d["loop"]["launches_per_eval"])
except Exception as e:
    print("$name failed", e)
PY
}
for l in 2 4 0 2; do
run cfg_l$l $l --cfg
run b16_l$l $l --batch 16
run r512_l$l $l --res 512
run interp_l$l $l --batch 4 --interp 11
done
