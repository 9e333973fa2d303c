#!/bin/bash
# experiment: cap the tuned split-K factors inside the real pipeline (two streams share the GPU)
for c in 0 1 2 4 6; do
  MKD_SPLITK_CAP=$c timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode 0 > gpurun_out/bench_sk_$c.log 2>&1
  echo "cap $c: $(tail -1 gpurun_out/bench_sk_$c.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],3), "img/s", round(d["loop"]["ms_per_eval"],3), "ms/eval", d["loop"]["launches_per_eval"], "launches")')"
done
