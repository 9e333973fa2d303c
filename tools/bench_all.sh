# every configuration BASELINE.md §4 quotes, one line each (name, images/s, ms per bench step, ms per evaluation, MFMA fraction of the loop)
mkdir -p gpurun_out
run() { name=$1; shift; python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/r4_bench_$name.json 2> gpurun_out/r4_bench_$name.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r4_bench_$name.json")); l=d["loop"]; print("$name", round(d["value"],2), round(d["ms_per_step"],1), round(l["ms_per_eval"],3), round(l["mfma_tflops_whole_loop"],1), round(l["mfma_frac_whole_loop"],4), l["launches_per_eval"])
except Exception as e:
    print("$name failed", e)
PY
}
run default
run latents --decode 0
run cfg --cfg
run r512 --res 512
run interp --batch 4 --interp 11
for b in 1 2 4 16; do run b$b --batch $b --decode 0; done
