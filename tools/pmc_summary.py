#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: mean counter value per kernel (last dispatch of each kernel name)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(agg):
    if not any(s in k for s in ("count_", "stats_", "em_acc", "bg_count")):
        continue
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("   %-28s last=%.4g  n=%d" % (c, v[-1], len(v)))
