#!/bin/bash
# usage (GPU box): tools/w12_size_sweep.sh -> W = 12 count kernels at several shard sizes, ns per 1000 windows per kernel.
# Premise check for "level-1 keys staged in the Infinity Cache": at 1/64 of a configs[3] shard the 32-bit keys of level 1
# (148 MB) are still in the 256 MB cache when count_rescatter12_kernel reads them; if that kernel is not markedly faster
# per window there than at full size (9.5 GB of keys through HBM), chunking the shard to fit the cache cannot pay.
export TMPDIR=/tmp
R=$PWD
for n in 195312 390625 781250 3125000 12500000; do
  D=$R/gpurun_out/w12sweep_$n
  cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --W 12 --nseq $n --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --k4-patterns 0 --em-stress-pwms 0 --config3-steps 0 > $D.log 2>&1
  cd $R
  python3 - $n $D <<'PY'
import csv, glob, sys
n = int(sys.argv[1]); win = n * 189
for f in glob.glob(sys.argv[2] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Name"]
        for key in ("count_scatter12", "count_rescatter12", "count_hist", "count_gather", "count_fixup"):
            if key in k:
                avg = float(r["AverageNs"])
                print("n=%9d %-22s avg %9.1f us  %7.3f ns per 1000 windows" % (n, key, avg / 1e3, avg / win * 1e3))
PY
  rm -rf $D $D.log
done
