# round 4: the plan switches that were only ever A/B'd at batch 8 256x256, at the big configurations:
# full-chip two-launch GroupNorm (MKD_GN_2K_MINHW), XCD ratio, slab-fed GroupNorm threshold, decoder lanes, LayerNorm on the fly.
mkdir -p gpurun_out
OUT=gpurun_out/exp_r4_bigcfg_switches.txt; : > $OUT
run() { env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA 2>gpurun_out/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$EXTRA | $* |', round(d['value'],3), 'img/s', round(d['loop']['ms_per_eval'],3), 'ms/eval')" >> $OUT || tail -5 gpurun_out/err.log >> $OUT; tail -1 $OUT; }
for EXTRA in "--res 512" "--cfg" "--batch 4 --interp 11"; do
  run MKD_X=0
  run MKD_GN_2K_MINHW=4096
  run MKD_GN_2K_MINHW=1024
  run MKD_XCD_AUTO_RATIO=1
  run MKD_GN_SLAB_MINC=640
  run MKD_GN_SLAB_MINC=320
  run MKD_DEC_LANES=0
  run MKD_LN_FLY=7
  run MKD_LN_FLY=0
  run MKD_X=0
done
