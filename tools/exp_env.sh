mkdir -p gpurun_out
run() { name=$1; shift; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA > gpurun_out/x_$name.json 2> gpurun_out/x_$name.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/x_$name.json")); k=d["kernel_classes_ms_per_eval"]
    print("$name", round(d["value"],2), round(d["loop"]["ms_per_eval"],3), d["loop"]["launches_per_eval"], "sum", round(sum(k.values()),3))
except Exception as e:
    print("$name failed", e)
PY
}
EXTRA="--graph 0"
run eager A=1
run mask1 MKD_CU_MASK=1
run mask2 MKD_CU_MASK=2
run mask1_nolanes MKD_CU_MASK=1 MKD_DEC_LANES=0
run mask1_helpers MKD_CU_MASK=1 MKD_LANE_HELPERS=1
run eager2 A=1
