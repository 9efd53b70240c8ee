#!/bin/bash
# usage (GPU box): tools/pmc_sweep.sh <pairs: 0 per-pattern kernel, 1 twin tiles> -> SQ / traffic counters of the sweep kernel (separate --pmc passes, --kernel-trace only)
R=$PWD; I=$1; O=$R/gpurun_out/pmc_sweep_$I; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
n=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAIT_ANY"; do
  n=$((n+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/p$n -- python3 $R/tools/sweep_probe.py --pairs $I --reps 3 --nseq 12500000 > $O/p$n.log 2>&1
done
cd $R; python3 - <<PY
import csv, glob, collections
for n in range(1, 5):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("$O/p%d/**/*counter_collection.csv" % n, recursive=True):
        for r in csv.DictReader(open(f)):
            if "stats" in r["Kernel_Name"]:
                a = agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, v in sorted(agg.items()):
        print("%-44s %-22s %14.0f per launch (%d launches)" % (k[0], k[1], v[0] / max(v[1], 1), v[1]))
PY
