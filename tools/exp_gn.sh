mkdir -p gpurun_out
run() { name=$1; shift; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > gpurun_out/e_$name.json 2> gpurun_out/e_$name.err; python - <<PY
import json
d=json.load(open("gpurun_out/e_$name.json")); k=d["kernel_classes_ms_per_eval"]
print("$name", round(d["value"],2), round(d["loop"]["ms_per_eval"],3), d["loop"]["launches_per_eval"], "gn", k["groupnorm"], "sum", round(sum(k.values()),3))
PY
}
EXTRA="" run fused1 MKD_GN_FUSED=1
EXTRA="" run fused0 MKD_GN_FUSED=0
EXTRA="" run fused2 MKD_GN_FUSED=2
EXTRA="--graph 0" run fused1_nograph MKD_GN_FUSED=1
EXTRA="--graph 0" run fused0_nograph MKD_GN_FUSED=0
EXTRA="" run fused1_nolanes MKD_GN_FUSED=1 MKD_DEC_LANES=0
EXTRA="" run fused0_nolanes MKD_GN_FUSED=0 MKD_DEC_LANES=0
