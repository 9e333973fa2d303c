# XCD-aware tile order applied to every GEMM / conv launch (MKD_XCD_MODE): correctness, then the loop
mkdir -p gpurun_out
for m in 1 2; do MKD_XCD_MODE=$m timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -m gpu -q -x 2>&1 | tail -2; done
run() { name=$1; shift; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA > gpurun_out/xcd_$name.json 2> gpurun_out/xcd_$name.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/xcd_$name.json")); k=d["kernel_classes_ms_per_eval"]; print("$name", round(d["value"],3), round(d["loop"]["ms_per_eval"],3), "serial sum", round(sum(k.values()),3))
except Exception as e:
    print("$name failed", e)
PY
}
for m in 0 1 2 0 1 2; do run m$m MKD_XCD_MODE=$m; done
EXTRA="--batch 1"; for m in 0 1 2; do run b1_m$m MKD_XCD_MODE=$m; done
EXTRA="--res 512"; for m in 0 1 2; do run r512_m$m MKD_XCD_MODE=$m; done
