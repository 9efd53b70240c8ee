#!/bin/bash
# usage (GPU box): tools/e2e_spread.sh [runs=8] [nseq=10000000] [W=10]
# The spread of peng_motif's wall time on the bench's FASTA, run after run, with the runtime start laps (PENGK_TIMING_CREATE)
# and the phase report (PENGK_TIMING): what the 100x end-to-end target is measured on (BASELINE configs[2]).
R=${1:-8}; N=${2:-10000000}; W=${3:-10}
D=/dev/shm/pengk_e2e_$$; mkdir -p $D
tools/synth_fasta $D/s.fa $N 200 1 0
for i in $(seq 1 $R); do
  t0=$(date +%s.%N)
  PENGK_TIMING=1 PENGK_TIMING_CREATE=1 peng-motif_amd/host/peng_motif $D/s.fa -w $W -o $D/o.meme -j $D/o.json > $D/out.txt 2> $D/err.txt
  t1=$(date +%s.%N)
  python3 -c "print(\"run $i: process wall %.3f s\" % ($t1 - $t0))"
  grep -E "pengk_create|timing" $D/err.txt | sed 's/^/    /'
done
rm -rf $D
