#!/bin/bash
# usage (GPU box): tools/e2e_gap.sh [runs=8] -> does one run's teardown in the driver slow the NEXT run's start?  peng_motif on
# the bench's FASTA back to back, then with half a second between the runs, then back to back again: wall, the runtime's start
# laps and the exit of every run.
R=${1:-8}
D=/dev/shm/pengk_e2e_$$; mkdir -p $D
tools/synth_fasta $D/s.fa 10000000 200 1 0
peng-motif_amd/host/peng_motif $D/s.fa -w 10 -o $D/o.meme > /dev/null 2>&1
for gap in 0 0.5 0 1.0; do
  echo "== $gap s between runs"
  for i in $(seq 1 $R); do
    sleep $gap
    t0=$(date +%s.%N)
    PENGK_TIMING=1 PENGK_TIMING_CREATE=1 peng-motif_amd/host/peng_motif $D/s.fa -w 10 -o $D/o.meme -j $D/o.json > /dev/null 2> $D/err.txt
    t1=$(date +%s.%N)
    python3 - $t0 $t1 $D/err.txt <<'PY'
import sys, re
t0, t1 = float(sys.argv[1]), float(sys.argv[2])
txt = open(sys.argv[3]).read()
g = lambda pat: float(re.search(pat, txt).group(1))
print("  wall %.3f  start %5.0f + %4.0f ms  main %.3f  exit %.3f" % (t1 - t0, g(r"runtime start\): ([0-9.]+)"), g(r"stream: ([0-9.]+)"), g(r"timing\] total: ([0-9.e-]+)"), t1 - t0 - g(r"timing\] total: ([0-9.e-]+)")))
PY
  done
done
rm -rf $D
