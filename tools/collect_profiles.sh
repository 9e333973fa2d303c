#!/bin/bash
# Run ON the GPU box (via gpurun): rocprofv3 kernel-trace stats of the bench command + the PMC passes (each in its own run,
# kernel-trace only beside --pmc), reduced to small per-kernel CSVs under gpurun_out/prof_summary/ (copy those into profiles/).
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_summary; mkdir -p $OUT
# provenance: gpurun snapshots carry no .git, so the caller passes the commit (MKD_HEAD=$(git rev-parse --short HEAD) gpurun ...)
python3 - <<PY > $OUT/r3_provenance.json
import json, time, hashlib
src = b''.join(open('makeupdiffuse_amd/csrc/' + f, 'rb').read() for f in ('engine.hip', 'kernels_gemm.hip', 'kernels_conv.hip', 'kernels_norm.hip', 'kernels_attn.hip', 'kernels_tfm.hip', 'kernels_misc.hip', 'gemm_tuned.inc'))
print(json.dumps({'commit': '${MKD_HEAD:-unknown}', 'utc': time.strftime('%Y-%m-%dT%H:%M:%SZ', time.gmtime()), 'csrc_sha256_16': hashlib.sha256(src).hexdigest()[:16],
                  'commands': 'tools/collect_profiles.sh (rocprofv3 --kernel-trace --stats of bench.py; one --pmc pass per counter set, kernel-trace only)'}))
PY
BENCH="bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph 0"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o r3 -- python3 $BENCH > $OUT/r3_rocprof_run.log 2>&1
F=$(find /tmp/prof_stats -name 'r3_kernel_stats.csv' | head -1); cp "$F" $OUT/r3_kernel_stats_bench_b8_256.csv
pmc() {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d /tmp/prof_$name -o r3 -- python3 bench.py --steps 1 --warmup 0 --ddim-steps 4 --no-cpu-baseline --graph 0 --decode 0 > /tmp/prof_$name.log 2>&1
  local D=$(dirname $(find /tmp/prof_$name -name 'r3_counter_collection.csv' | head -1))
  python3 tools/summarize_prof.py "$D" r3 --out $OUT/r3_pmc_$name.csv
  echo "pmc $name done"
}
pmc fetch_size_kb FETCH_SIZE
pmc write_size_kb WRITE_SIZE
pmc mfma_busy SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
pmc l2_hit TCC_HIT_sum TCC_MISS_sum
pmc lds_wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
# stream timeline of the eager 2-stream loop: queue busy / overlap / gaps (tools/timeline.py)
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o r3 -- python3 bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --graph 0 --decode 0 > /tmp/prof_tl.log 2>&1
python3 tools/timeline.py $(find /tmp/prof_tl -name 'r3_kernel_trace.csv' | head -1) --out $OUT/r3_timeline_eager_b8_256.txt > /dev/null
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph 0 --decode 0 --ops-csv $OUT/r3_ops_per_launch_b8_256.csv > /dev/null 2>&1
ls -la $OUT
