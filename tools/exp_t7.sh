# A/B of two builds over the secondary configurations: libmkd_base.so (previous table) vs libmkd.so
mkdir -p gpurun_out
run() { name=$1; shift; lib=$1; shift; env MKD_LIB_PATH=$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/t7_$name.json 2> gpurun_out/t7_$name.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/t7_$name.json")); print("$name", round(d["value"],3), round(d["ms_per_step"],3))
except Exception as e:
    print("$name failed", e)
PY
}
B=$PWD/makeupdiffuse_amd/libmkd_base.so; N=$PWD/makeupdiffuse_amd/libmkd.so
for b in 1 2 4 16; do run b${b}_base $B --batch $b --decode 0; run b${b}_new $N --batch $b --decode 0; done
run b8_base $B; run b8_new $N
run cfg_base $B --cfg; run cfg_new $N --cfg
run r512_base $B --res 512; run r512_new $N --res 512
run interp_base $B --batch 4 --interp 11; run interp_new $N --batch 4 --interp 11
run b8_base2 $B; run b8_new2 $N
