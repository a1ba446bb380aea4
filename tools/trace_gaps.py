#!/usr/bin/env python
"""Idle time around every kernel of a rocprofv3 --kernel-trace run: for each kernel name the average gap between the end of
the previous kernel on the device and its own start ("before") and between its end and the next kernel's start ("after"),
beside its average duration.  A kernel that is fast in isolation but delays its neighbours shows up here, not in --stats.
usage: python tools/trace_gaps.py <dir with *kernel_trace.csv> [min_calls]"""
import collections
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 50
acc = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for i, (s, e, k) in enumerate(rows):
    a = acc[k]
    a[0] += 1
    a[1] += (e - s) / 1e3
    if i > 0:
        a[2] += max(0, s - rows[i - 1][1]) / 1e3
    if i + 1 < len(rows):
        a[3] += max(0, rows[i + 1][0] - e) / 1e3
span = (rows[-1][1] - rows[0][0]) / 1e3
busy = sum((e - s) for s, e, _ in rows) / 1e3
print(f"{len(rows)} launches, span {span / 1e3:.1f} ms, sum of durations {busy / 1e3:.1f} ms")
print(f"{'kernel':70s} {'calls':>7s} {'avg us':>8s} {'gap before':>10s} {'gap after':>10s}")
for k, (n, d, gb, ga) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if n >= min_calls:
        name = k.replace("(anonymous namespace)::", "")[:70]
        print(f"{name:70s} {n:7d} {d / n:8.2f} {gb / n:10.2f} {ga / n:10.2f}")
