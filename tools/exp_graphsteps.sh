# k DDIM steps per captured graph (MKD_GRAPH_STEPS): is there a gap between graph launches?
mkdir -p gpurun_out
run() { name=$1; shift; b=$1; shift; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 --batch $b > gpurun_out/gs_$name.json 2> gpurun_out/gs_$name.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/gs_$name.json")); print("$name", round(d["value"],3), round(d["loop"]["ms_per_eval"],3))
except Exception as e:
    print("$name failed", e)
PY
}
for b in 8 1; do for k in 1 2 5 10 25 1; do run b${b}_k$k $b MKD_GRAPH_STEPS=$k; done; done
