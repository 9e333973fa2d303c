# A/B two builds of libmkd on one box: MKD_LIB_PATH=<old> vs the in-tree library, alternating
for i in 1 2 3; do for v in old new; do
if [ $v = old ]; then export MKD_LIB_PATH=$PWD/makeupdiffuse_amd/libmkd_old.so; else unset MKD_LIB_PATH; fi
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --graph 0 --decode 0 $EXTRA 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],3), round(d['loop']['ms_per_eval'],4))" >> gpurun_out/ab.log || exit 1
done; done
