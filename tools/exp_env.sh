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
# usage (on the GPU box): edit the legs below; each leg = a name + environment assignments for one bench run
run base A=1
run ln_fly0 MKD_LN_FLY=0
run ln_fly3 MKD_LN_FLY=3
run ln_fly7 MKD_LN_FLY=7
run base2 A=1
