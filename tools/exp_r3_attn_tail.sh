# round 3: the 16-deep tail MFMA for d_head 40 / 80 with an accumulator of its own (makeupdiffuse_amd/libmkd_tail.so, built with
# tools/build_variant.sh tail -DMKD_ATTN_TAIL=1) against the default (heads padded to the 32-deep step): parity first, then time
mkdir -p gpurun_out
MKD_LIB_PATH=$PWD/makeupdiffuse_amd/libmkd_tail.so timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k attention 2>&1 | tail -3
for i in 1 2; do
  python tools/bench_attn.py 2>/dev/null | tail -10 | sed 's/^/default /'
  MKD_LIB_PATH=$PWD/makeupdiffuse_amd/libmkd_tail.so python tools/bench_attn.py 2>/dev/null | tail -10 | sed 's/^/tail    /'
done
for i in 1 2; do for v in default tail; do
if [ $v = tail ]; then export MKD_LIB_PATH=$PWD/makeupdiffuse_amd/libmkd_tail.so; else unset MKD_LIB_PATH; fi
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_classes_ms_per_eval']; print('$v', round(d['value'],3), 'ms/eval', round(d['loop']['ms_per_eval'],3), 'attention serial ms', k.get('attention'))"
done; done
