# small-batch defaults: LayerNorm-on-the-fly mask, slab-fed GroupNorm threshold and producer GroupNorm statistics at batch 1 / 2 / 4
mkdir -p gpurun_out
run() { name=$1; shift; b=$1; shift; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 --batch $b > gpurun_out/sb_$name.json 2> gpurun_out/sb_$name.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/sb_$name.json")); print("$name", round(d["value"],3), round(d["loop"]["ms_per_eval"],3), d["loop"]["launches_per_eval"])
except Exception as e:
    print("$name failed", e)
PY
}
for b in 1 2 4; do
run b${b}_base $b A=1
run b${b}_fly3 $b MKD_LN_FLY=3
run b${b}_fly7 $b MKD_LN_FLY=7
run b${b}_slab320 $b MKD_GN_SLAB_MINC=320
run b${b}_fly7_slab320 $b MKD_LN_FLY=7 MKD_GN_SLAB_MINC=320
run b${b}_gnfused $b MKD_GN_FUSED=1
run b${b}_base2 $b A=1
done
