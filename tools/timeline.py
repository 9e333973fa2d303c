#!/usr/bin/env python
"""Timeline analysis of a rocprofv3 --kernel-trace CSV of bench.py: per HIP stream (queue) busy time, idle gaps, overlap between
streams, and the kernels around the largest gaps.  Answers "is the sampling loop bound by kernel execution or by dependencies /
launch gaps?" - the question a per-kernel statistics table cannot answer for a multi-stream plan.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o r2 -- python3 bench.py --steps 1 --warmup 1 --ddim-steps 10 --graph 0 ...
    python tools/timeline.py /tmp/kt/.../r2_kernel_trace.csv [--skip-frac 0.5]
"""
import argparse
import collections
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--steps', type=int, default=4, help='DDIM steps (from the end of the trace) in the analysed window')
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    rows = list(csv.DictReader(open(args.csv)))
    ev = []
    for r in rows:
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '0'), r['Kernel_Name']))
    ev.sort()
    # the steady sampling loop: whole DDIM steps between the end of the (n+1)-th last and the end of the last ddim_step kernel
    steps = [e for e in ev if 'ddim_step' in e[3]]
    nwin = max(1, min(args.steps, len(steps) - 1))
    lo, hi = steps[-1 - nwin][1], steps[-1][1]
    ev = [e for e in ev if lo <= e[0] and e[1] <= hi]
    span = hi - lo
    byq = collections.defaultdict(list)
    for e in ev:
        byq[e[2]].append(e)
    lines = [f'window {span / 1e6:.3f} ms = {nwin} DDIM steps ({span / 1e6 / nwin:.3f} ms per step), {len(ev)} kernels, {len(byq)} queues']
    # union busy time (any queue), and time with >= 2 kernels running
    pts = []
    for s, e, q, n in ev:
        pts.append((s, 1)); pts.append((e, -1))
    pts.sort()
    depth = 0; last = lo; busy1 = busy2 = 0
    for t, d in pts:
        if depth >= 1: busy1 += t - last
        if depth >= 2: busy2 += t - last
        depth += d; last = t
    lines.append(f'some kernel running {100 * busy1 / span:.1f} % of the window; >= 2 kernels concurrently {100 * busy2 / span:.1f} %; GPU idle {100 * (1 - busy1 / span):.1f} %')
    for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        lst.sort()
        busy = sum(e - s for s, e, _, _ in lst)
        gaps = [(lst[i + 1][0] - lst[i][1], lst[i][3], lst[i + 1][3]) for i in range(len(lst) - 1)]
        pos = [g for g in gaps if g[0] > 0]
        gsum = sum(g[0] for g in pos)
        small = sum(g[0] for g in pos if g[0] < 5000)
        lines.append(f'queue {q}: {len(lst)} kernels, busy {busy / 1e6:.3f} ms ({100 * busy / span:.1f} %), avg kernel {busy / len(lst) / 1e3:.2f} us, '
                     f'gaps {gsum / 1e6:.3f} ms (avg {gsum / max(1, len(pos)) / 1e3:.2f} us; gaps < 5 us sum {small / 1e6:.3f} ms)')
        big = sorted(pos, key=lambda g: -g[0])[:6]
        for g in big:
            lines.append(f'    gap {g[0] / 1e3:.1f} us after {g[1][:60]} before {g[2][:60]}')
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, q, n in ev:
        k = n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:70]
        agg[k][0] += 1; agg[k][1] += e - s
    lines.append('kernel time in the window (concurrent execution):')
    tot = sum(v[1] for v in agg.values())
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        lines.append(f'  {v[1] / 1e6:8.3f} ms {100 * v[1] / tot:5.1f} %  n={v[0]:5d} avg {v[1] / v[0] / 1e3:7.2f} us  {k}')
    lines.append(f'  total kernel time {tot / 1e6:.3f} ms = {tot / span:.2f} x the window')
    txt = '\n'.join(lines)
    print(txt)
    if args.out:
        open(args.out, 'w').write(txt + '\n')


if __name__ == '__main__':
    main()
