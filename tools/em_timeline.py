#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) and prints, for the em_* kernels, the average duration per
kernel and the average GAP between the end of one kernel and the start of the next on the same queue -- what a dependent
launch costs on top of the kernels themselves.   python tools/em_timeline.py <dir with the trace>"""
import collections
import csv
import glob
import re
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
byq = collections.defaultdict(list)
for s, e, k, q in rows:
    byq[q].append((s, e, k))
dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
for q, v in byq.items():
    for i, (s, e, k) in enumerate(v):
        if "em_" not in k:
            continue
        name = re.search(r"(em_\w+)", k).group(1)
        dur[name].append(e - s)
        if i and "em_" in v[i - 1][2]:
            gap[name].append(s - v[i - 1][1])
for name in sorted(dur):
    d, g = dur[name], gap.get(name, [0])
    print("%-40s n=%5d  avg %8.2f us   gap before it: avg %6.2f us  median %6.2f us" % (
        name, len(d), sum(d) / len(d) / 1e3, sum(g) / len(g) / 1e3, sorted(g)[len(g) // 2] / 1e3))
