# round 3: kernel timeline of the GRAPH-replayed loop (batch 8, latents only): how much of a step has no kernel running at all, where
# the largest holes are (phase changes of a step), concurrency of the queues

export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tlg -o tlg -- python3 bench.py --steps 1 --warmup 1 --ddim-steps 20 --no-cpu-baseline --decode 0 > gpurun_out/prof_tlg.log 2>&1 || tail -20 gpurun_out/prof_tlg.log
python3 tools/timeline.py $(find /tmp/prof_tlg -name 'tlg_kernel_trace.csv' | head -1) --steps 8 --out gpurun_out/r3_timeline_graph_b8_256.txt > /dev/null
head -40 gpurun_out/r3_timeline_graph_b8_256.txt
