# round 3: graph replay forms under the final plan (two chains + time-embedding table): one captured graph per 5 steps (default),
# linear graphs per stream (MKD_GRAPH_MODE=2), other step counts per graph, eager launches
mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA 2>gpurun_out/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$* $EXTRA', round(d['value'],3), 'ms/eval', round(d['loop']['ms_per_eval'],3))" || tail -5 gpurun_out/err.log; }
for i in 1 2; do
  run MKD_X=0
  run MKD_GRAPH_MODE=2
  run MKD_GRAPH_STEPS=1
  run MKD_GRAPH_STEPS=10
  run MKD_GRAPH_STEPS=25
  EXTRA="--graph 0" run MKD_X=0
done
