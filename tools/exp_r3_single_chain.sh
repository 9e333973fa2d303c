# round 3: the decisions that were taken under two contending chains (LayerNorm fusion, slab-fed GroupNorm, the tuned table, graph
# mode), measured again under the single grouped encoder chain (MKD_ENC_GROUP=1); first / last line = the two-chain default
mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA 2>gpurun_out/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_classes_ms_per_eval']; print('$*', round(d['value'],3), 'ms/eval', round(d['loop']['ms_per_eval'],3), 'launches', d['loop']['launches_per_eval'], 'serial sum', round(sum(k.values()),3))" || tail -5 gpurun_out/err.log; }
run MKD_ENC_GROUP=0
run MKD_ENC_GROUP=1
run MKD_ENC_GROUP=1 MKD_FUSE_LN=1
run MKD_ENC_GROUP=1 MKD_LN_FLY=7
run MKD_ENC_GROUP=1 MKD_LN_FLY=0
run MKD_ENC_GROUP=1 MKD_GN_SLAB_MINC=320
run MKD_ENC_GROUP=1 MKD_GN_SLAB=0
run MKD_ENC_GROUP=1 MKD_NO_TABLE=1
run MKD_ENC_GROUP=1 MKD_GRAPH_MODE=2
run MKD_ENC_GROUP=1 MKD_DEC_LANES=4
run MKD_ENC_GROUP=1 MKD_GN_2K_MINHW=1024
run MKD_ENC_GROUP=0
