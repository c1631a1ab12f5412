#!/usr/bin/env python
"""Kernel timeline of the last frame of a `rocprofv3 --kernel-trace` run: start, end and gap to the previous kernel's end, per launch.
usage: scripts/kernel_timeline.py <dir with *_kernel_trace.csv> [n_last_kernels]"""
import csv
import glob
import sys

d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48], r.get("Queue_Id", "")))
rows.sort()
rows = rows[-n_last:]
t0 = rows[0][0]
prev_end = t0
for s, e, k, q in rows:
    print("%9.1f us  +%8.1f us  gap %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, k))
    prev_end = max(prev_end, e)
print("span %.1f us, kernels busy (union) ..." % ((rows[-1][1] - t0) / 1e3))
