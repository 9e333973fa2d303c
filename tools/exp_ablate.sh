# Ablation bound: the loop with a whole kernel class left out (experiment build libmkd_ablate.so = -DMKD_EXP_ABLATE, results are WRONG on
# purpose).  What the loop gains when a class costs nothing at all bounds every optimisation of that class.  EXTRA = extra bench flags
mkdir -p gpurun_out
D=$PWD/makeupdiffuse_amd
run() { name=$1; shift; env MKD_BENCH_ALLOW_NONFINITE=1 MKD_LIB_PATH=$D/libmkd_ablate.so "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],3), round(d['loop']['ms_per_eval'],3))"; }
run base MKD_EXP_SKIP=0
run no_groupnorm MKD_EXP_SKIP=1
run no_layernorm MKD_EXP_SKIP=2
run no_attention MKD_EXP_SKIP=4
run no_splitk_reduce MKD_EXP_SKIP=8
run no_gn_ln_reduce MKD_EXP_SKIP=11
run none_of_them MKD_EXP_SKIP=15
run base2 MKD_EXP_SKIP=0
# the launches stay, only the work goes (a one-element fill kernel in place of every kernel of the class; 16 = every GEMM / convolution)
run empty_groupnorm MKD_EXP_EMPTY=1
run empty_layernorm MKD_EXP_EMPTY=2
run empty_attention MKD_EXP_EMPTY=4
run empty_gemm MKD_EXP_EMPTY=16
run empty_all MKD_EXP_EMPTY=23
run base3 MKD_EXP_SKIP=0
