#!/bin/bash
# usage (GPU box): tools/pmc_em.sh <tag>  -- memory-side counters of the serial EM's kernels (one bench step), per kernel
export TMPDIR=/tmp
R=$PWD; T=$1; OUT=$R/gpurun_out/$T; mkdir -p $OUT
cd /tmp
i=0
for pmc in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --k4-patterns 0 --em-stress-pwms 0 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
cd $R
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "em_" not in k: continue
        acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k, {c: "%.4g (n=%d)" % (sum(v) / len(v), len(v)) for c, v in d.items()})
PY
