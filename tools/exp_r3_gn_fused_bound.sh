# round 3: the producer-statistics GroupNorm (MKD_GN_FUSED=1) today, and its bound without the device-scope atomics (experiment build
# -DMKD_EXP_NO_FLUSH: the tile's LDS accumulator is never flushed - WRONG statistics, only the time is read)
mkdir -p gpurun_out
D=$PWD/makeupdiffuse_amd
run() { name=$1; shift; env MKD_BENCH_ALLOW_NONFINITE=1 "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 2>gpurun_out/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],3), 'ms/eval', round(d['loop']['ms_per_eval'],3), 'launches', d['loop']['launches_per_eval'])" || tail -3 gpurun_out/err.log; }
for i in 1 2 3; do
  run base A=1
  run gn_fused_atomics MKD_GN_FUSED=1
  run gn_fused_no_flush MKD_GN_FUSED=1 MKD_LIB_PATH=$D/libmkd_noflush.so
done
