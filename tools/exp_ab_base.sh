# A/B: makeupdiffuse_amd/libmkd_base.so (previous build) vs the in-tree library, alternating; EXTRA = extra bench flags
mkdir -p gpurun_out
for i in 1 2 3; do for v in base new; do
if [ $v = base ]; then export MKD_LIB_PATH=$PWD/makeupdiffuse_amd/libmkd_base.so; else unset MKD_LIB_PATH; fi
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_classes_ms_per_eval']; print('$v', round(d['value'],3), round(d['loop']['ms_per_eval'],3), 'serial sum', round(sum(k.values()),3))"
done; done
