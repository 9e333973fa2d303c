#!/usr/bin/env python
"""Reduces rocprofv3 CSV output (kernel trace / counter collection) to per-kernel summaries small enough to commit.

    python tools/summarize_prof.py <dir> <prefix> [--out profiles/xyz.csv]
Handles: <prefix>_kernel_stats.csv (copied as is), <prefix>_counter_collection.csv (+ matching _kernel_trace.csv)."""
import collections
import csv
import os
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return name.split('(')[0]


def main():
    d, prefix = sys.argv[1], sys.argv[2]
    out = sys.argv[sys.argv.index('--out') + 1] if '--out' in sys.argv else os.path.join(d, prefix + '_summary.csv')
    cc = os.path.join(d, prefix + '_counter_collection.csv')
    kt = os.path.join(d, prefix + '_kernel_trace.csv')
    dur = {}
    if os.path.exists(kt):
        for r in csv.DictReader(open(kt)):
            dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    counters = set()
    for r in csv.DictReader(open(cc)):
        k = short(r['Kernel_Name'])
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        counters.add(r['Counter_Name'])
        calls[k].add(r['Dispatch_Id'])
    counters = sorted(counters)
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'dispatches', 'avg_us'] + [f'avg_{c}' for c in counters])
        for k in sorted(agg, key=lambda k: -sum(dur.get(i, 0) for i in calls[k])):
            n = len(calls[k])
            avg_us = sum(dur.get(i, 0) for i in calls[k]) / n if dur else ''
            w.writerow([k, n, f'{avg_us:.3f}' if dur else ''] + [f'{agg[k][c] / n:.3f}' for c in counters])
    print('wrote', out)


if __name__ == '__main__':
    main()
