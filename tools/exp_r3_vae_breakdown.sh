# round 3: where the first-stage decode of a batch (8 x 256x256) spends its time: rocprofv3 kernel statistics of tools/bench_vae.py
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vae -o vae -- python3 tools/bench_vae.py > gpurun_out/vae_run.log 2>&1
tail -3 gpurun_out/vae_run.log
python3 - <<PY
import csv, glob
f = glob.glob('/tmp/prof_vae/**/vae_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:22]:
    print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
