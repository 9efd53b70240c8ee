#!/bin/bash
# usage (GPU box): PENGK_W12_CHUNKS=4 tools/w12_timeline.sh -> the count kernels of the last W = 12 step as a timeline with their queues
R=$PWD; D=$R/gpurun_out/w12tl_$$
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --W 12 --nseq 12500000 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --k4-patterns 0 --config3-steps 0 --em-stress-pwms 0 --pipelined 0 --strong 0 --pwms 0 > $D.log 2>&1
cd $R; python3 - <<PY
import csv, glob, re
rows = []
for f in glob.glob("$D/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "count_" in r["Kernel_Name"] or "bg_finish" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), re.search(r"((count|bg)_\w+)", r["Kernel_Name"]).group(1)))
rows.sort()
last = max(i for i, r in enumerate(rows) if r[3].startswith("count_scatter12") and (i == 0 or not rows[i - 1][3].startswith("count_")) or (r[3].startswith("count_scatter12") and i > 0 and rows[i-1][3] in ("count_fixup_kernel",)))
rows = rows[last:]
t0 = rows[0][0]
for s, e, q, k in rows[:40]:
    print("%9.1f %9.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, q, k))
PY
rm -rf $D $D.log
