D=$PWD/makeupdiffuse_amd
run() { name=$1; shift; env MKD_BENCH_ALLOW_NONFINITE=1 MKD_LIB_PATH=$D/libmkd_ablate.so "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],3), round(d['loop']['ms_per_eval'],3))"; }
run mode1_base MKD_GRAPH_MODE=1
run mode2_base MKD_GRAPH_MODE=2
run mode1_empty_all MKD_GRAPH_MODE=1 MKD_EXP_EMPTY=23
run mode2_empty_all MKD_GRAPH_MODE=2 MKD_EXP_EMPTY=23
run mode1_empty_gemm MKD_GRAPH_MODE=1 MKD_EXP_EMPTY=16
run mode2_empty_gemm MKD_GRAPH_MODE=2 MKD_EXP_EMPTY=16
EXTRA="--graph 0" run eager_empty_all MKD_EXP_EMPTY=23
