# Bound for folding the ResBlocks' 1x1 skip_connection GEMM into conv2 (priced, never measured): the loop with
# those GEMMs left out (experiment build libmkd_ablate.so = tools/build_variant.sh ablate -DMKD_EXP_ABLATE; results are WRONG on purpose)
mkdir -p gpurun_out
D=$PWD/makeupdiffuse_amd
run() { name=$1; shift; env MKD_BENCH_ALLOW_NONFINITE=1 MKD_LIB_PATH=$D/libmkd_ablate.so "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --live-pmc 0 --decode 0 $EXTRA 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$EXTRA', '$name', round(d['value'],3), round(d['loop']['ms_per_eval'],4), d['loop']['launches_per_eval'])"; }
for i in 1 2; do run base MKD_EXP_SKIP=0; run no_skip_gemms MKD_EXP_SKIP=32; done
