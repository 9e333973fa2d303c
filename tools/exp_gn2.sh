mkdir -p gpurun_out
run() { name=$1; shift; env MKD_BENCH_ALLOW_NONFINITE=1 "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/e_$name.json 2> gpurun_out/e_$name.err; python - <<PY
import json
d=json.load(open("gpurun_out/e_$name.json")); k=d["kernel_classes_ms_per_eval"]
print("$name", round(d["value"],2), round(d["loop"]["ms_per_eval"],3), d["loop"]["launches_per_eval"], "gn", k["groupnorm"], "sum", round(sum(k.values()),3), "patch128x64", k.get("gemm_conv3x3_patch128x64"), "c64x160s2", k.get("gemm_conv3x3_64x160_s2"), "c64x128", k.get("gemm_conv3x3_64x128"))
PY
}
D=$PWD/makeupdiffuse_amd
run fused0 MKD_GN_FUSED=0
run fused1 MKD_GN_FUSED=1
run fused1_noflush MKD_GN_FUSED=1 MKD_LIB_PATH=$D/libmkd_noflush.so
run fused1_nowave MKD_GN_FUSED=1 MKD_LIB_PATH=$D/libmkd_nowave.so
run fused0b MKD_GN_FUSED=0
