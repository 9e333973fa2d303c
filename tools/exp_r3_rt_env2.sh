# round 3: the runtime switches round 2's sweep (exp_rt_env.sh) left out - where kernel arguments live and how signals are waited for
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA 2>gpurun_out/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],2), round(d['loop']['ms_per_eval'],3))" || echo "$name failed: $(tail -2 gpurun_out/err.log)"; }
for i in 1; do
run base A=1
run dev_kernarg0 HIP_FORCE_DEV_KERNARG=0
run dev_kernarg1 HIP_FORCE_DEV_KERNARG=1
run kernarg_copy_opt0 DEBUG_HIP_KERNARG_COPY_OPT=0
run kernarg_copy_opt1 DEBUG_HIP_KERNARG_COPY_OPT=1
run fgs_kernarg0 ROC_USE_FGS_KERNARG=0
run fgs_kernarg1 ROC_USE_FGS_KERNARG=1
run hdp_flush_wa0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run hdp_flush_wa1 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1
run streamops_cp_wait1 GPU_STREAMOPS_CP_WAIT=1
run streamops_cp_wait0 GPU_STREAMOPS_CP_WAIT=0
# (ROC_SYSTEM_SCOPE_SIGNAL=0 is NOT in the sweep: the graph replay never completes with it - the run was killed after 7 silent minutes)
run compute_rings4 GPU_NUM_COMPUTE_RINGS=4
run max_batch1 DEBUG_CLR_MAX_BATCH_SIZE=1
done
