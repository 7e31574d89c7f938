#!/usr/bin/env python3
"""Durations and inter-kernel gaps of the last steps in a rocprofv3 kernel trace CSV (tools/prof_bench.sh output)."""
import csv, glob, os, sys
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, '**', '*_kernel_trace.csv'), recursive=True), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
prev = None
agg = {}
for r in rows[-120:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name']
    name = 'play' if 'k_td_play' in name else 'owner' if 'update_owner' in name else 'apply' if 'k_apply' in name else name[:24]
    if prev is not None:
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1; a[1] += (e - s) / 1e3; a[2] += (s - prev) / 1e3
    prev = e
for k, (n, dur, gap) in agg.items():
    print(f'{k:26s} n {n:3d}  mean duration {dur / n:8.1f} us  mean gap before {gap / n:6.1f} us  grid {""}')
