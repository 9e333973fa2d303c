# round 3: what each kernel class costs the loop when a step is replayed as LINEAR graphs per stream (MKD_GRAPH_MODE=2: the dispatcher
# term drops from 1.65 to 0.93 ms per evaluation) instead of one captured graph with branches (mode 1, default).  Ablation build.
mkdir -p gpurun_out
D=$PWD/makeupdiffuse_amd
run() { name=$1; shift; env MKD_BENCH_ALLOW_NONFINITE=1 MKD_LIB_PATH=$D/libmkd_ablate.so "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 --live-pmc 0 $EXTRA 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],3), round(d['loop']['ms_per_eval'],3))"; }
for m in 1 2; do
  run mode${m}_base MKD_GRAPH_MODE=$m
  run mode${m}_no_groupnorm MKD_GRAPH_MODE=$m MKD_EXP_SKIP=1
  run mode${m}_no_layernorm MKD_GRAPH_MODE=$m MKD_EXP_SKIP=2
  run mode${m}_no_attention MKD_GRAPH_MODE=$m MKD_EXP_SKIP=4
  run mode${m}_no_reduce MKD_GRAPH_MODE=$m MKD_EXP_SKIP=8
  run mode${m}_none_of_them MKD_GRAPH_MODE=$m MKD_EXP_SKIP=15
  run mode${m}_empty_groupnorm MKD_GRAPH_MODE=$m MKD_EXP_EMPTY=1
  run mode${m}_empty_attention MKD_GRAPH_MODE=$m MKD_EXP_EMPTY=4
  run mode${m}_empty_gemm MKD_GRAPH_MODE=$m MKD_EXP_EMPTY=16
  run mode${m}_empty_all MKD_GRAPH_MODE=$m MKD_EXP_EMPTY=23
done
