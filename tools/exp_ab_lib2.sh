# A/B of builds (MKD_LIB_PATH) in the sampling loop, alternating on one box: LIBS="a.so b.so" EXTRA="--res 512" ROUNDS=2
mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA 2>gpurun_out/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$EXTRA | $* |', round(d['value'],3), 'img/s', round(d['loop']['ms_per_eval'],3), 'ms/eval')" || tail -5 gpurun_out/err.log; }
for i in $(seq 1 ${ROUNDS:-2}); do for l in $LIBS; do run MKD_LIB_PATH=makeupdiffuse_amd/$l; done; done
