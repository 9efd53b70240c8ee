#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and prints the em_* kernels of the LAST pengk_em call as a timeline: start and end
in us since the call's first kernel, queue, kernel.   python tools/em_rawtrace.py <dir with the trace> [max lines]"""
import csv
import glob
import re
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "em_" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), re.search(r"(em_\w+)", r["Kernel_Name"]).group(1)))
rows.sort()
last = max(i for i, r in enumerate(rows) if r[3] in ("em_init_kernel", "em_serial_setup_kernel"))  # (a call's first kernel)
rows = rows[last:]
t0 = rows[0][0]
for s, e, q, k in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 80]:
    print("%9.1f %9.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, q, k))
