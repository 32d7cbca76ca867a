#!/usr/bin/env python3
"""Gaps between consecutive kernels of one stream in a rocprofv3 --kernel-trace CSV (development tool): for every kernel whose
name contains one of the given substrings, the idle time in front of it and its duration, averaged.
usage: trace_gaps.py <kernel_trace.csv> substring [substring ...]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
keys = sys.argv[2:]
acc = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    for k in keys:
        if k in n1:
            prev = next((kk for kk in keys if kk in n0), "other")
            a = acc[(prev, k)]
            a[0] += 1
            a[1] += max(0, s1 - e0)
            a[2] += e1 - s1
            a[3] = max(a[3], s1 - e0)
for (prev, k), (c, gap, dur, mx) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{prev:28s} -> {k:28s} x{c:6d}: gap in front {gap / c / 1e3:7.2f} us (max {mx / 1e3:8.1f}), duration {dur / c / 1e3:7.2f} us")
