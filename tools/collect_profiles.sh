#!/bin/bash
# Run ON the GPU box (via gpurun): rocprofv3 kernel-trace stats of the bench command + the PMC passes (each in its own run,
# kernel-trace only beside --pmc), reduced to small per-kernel CSVs under gpurun_out/prof_summary/ (copy those into profiles/).
#   TAG=b8_256 EXTRA=""                     bash tools/collect_profiles.sh      (the default bench configuration)
#   TAG=r512   EXTRA="--res 512"            bash tools/collect_profiles.sh
#   TAG=cfg    EXTRA="--cfg"                bash tools/collect_profiles.sh
#   TAG=interp EXTRA="--batch 4 --interp 11" bash tools/collect_profiles.sh
# ROUND (default r4) prefixes the file names.  LIGHT=1: kernel stats + FETCH/WRITE/L2 passes only (no timeline / lds_wait / ops csv).
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=${ROUND:-r4}; TAG=${TAG:-b8_256}
OUT=gpurun_out/prof_summary; mkdir -p $OUT
# provenance: gpurun snapshots carry no .git, so the caller passes the commit (MKD_HEAD=$(git rev-parse --short HEAD) gpurun ...)
python3 - <<PY > $OUT/${R}_provenance_${TAG}.json
import json, time, hashlib
src = b''.join(open('makeupdiffuse_amd/csrc/' + f, 'rb').read() for f in ('engine.hip', 'kernels_gemm.hip', 'kernels_conv.hip', 'kernels_norm.hip', 'kernels_attn.hip', 'kernels_tfm.hip', 'kernels_misc.hip', 'gemm_tuned.inc'))
print(json.dumps({'commit': '${MKD_HEAD:-unknown}', 'utc': time.strftime('%Y-%m-%dT%H:%M:%SZ', time.gmtime()), 'csrc_sha256_16': hashlib.sha256(src).hexdigest()[:16],
                  'bench_args': '${EXTRA}', 'commands': 'tools/collect_profiles.sh (rocprofv3 --kernel-trace --stats of bench.py; one --pmc pass per counter set, kernel-trace only)'}))
PY
BENCH="bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph 0 $EXTRA"
rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o p -- python3 $BENCH > $OUT/${R}_rocprof_run_${TAG}.log 2>&1
F=$(find /tmp/prof_stats -name 'p_kernel_stats.csv' | head -1); cp "$F" $OUT/${R}_kernel_stats_bench_${TAG}.csv
grep "^{" $OUT/${R}_rocprof_run_${TAG}.log | tail -1 > $OUT/${R}_bench_line_under_profiler_${TAG}.json || true
pmc() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d /tmp/prof_$name -o p -- python3 bench.py --steps 1 --warmup 0 --ddim-steps 4 --no-cpu-baseline --graph 0 --decode 0 $EXTRA > /tmp/prof_$name.log 2>&1
  local D=$(dirname $(find /tmp/prof_$name -name 'p_counter_collection.csv' | head -1))
  python3 tools/summarize_prof.py "$D" p --out $OUT/${R}_pmc_${name}_${TAG}.csv
  echo "pmc $name done"
}
pmc fetch_size_kb FETCH_SIZE
pmc write_size_kb WRITE_SIZE
pmc l2_hit TCC_HIT_sum TCC_MISS_sum
pmc mfma_busy SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
if [ -z "$LIGHT" ]; then
  pmc lds_wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  # stream timeline of the eager 2-stream loop: queue busy / overlap / gaps (tools/timeline.py)
  rm -rf /tmp/prof_tl
  rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o p -- python3 bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --graph 0 --decode 0 $EXTRA > /tmp/prof_tl.log 2>&1
  python3 tools/timeline.py $(find /tmp/prof_tl -name 'p_kernel_trace.csv' | head -1) --out $OUT/${R}_timeline_eager_${TAG}.txt > /dev/null
  python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph 0 --decode 0 $EXTRA --ops-csv $OUT/${R}_ops_per_launch_${TAG}.csv > /dev/null 2>&1
fi
ls -la $OUT
