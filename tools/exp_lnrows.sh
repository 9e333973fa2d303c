# row threshold below which a transformer takes all three LayerNorms on the fly (MKD_LN_FLY_ROWS)
mkdir -p gpurun_out
run() { name=$1; shift; b=$1; shift; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 --batch $b > gpurun_out/lr_$name.json 2> gpurun_out/lr_$name.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/lr_$name.json")); print("$name", round(d["value"],3), round(d["loop"]["ms_per_eval"],3), d["loop"]["launches_per_eval"])
except Exception as e:
    print("$name failed", e)
PY
}
for b in 8 1 2 4 16; do
for r in 0 128 256 512 1024 2048 4096 0; do run b${b}_rows$r $b MKD_LN_FLY_ROWS=$r; done
done
