#!/bin/bash
# usage (GPU box): tools/e2e_timing.sh [nseq] [W]  -> PENGK_TIMING phase report of peng_motif on the synthetic FASTA, three runs
N=${1:-10000000}; W=${2:-10}
D=/dev/shm/pengk_e2e_$$; mkdir -p $D
tools/synth_fasta $D/s.fa $N 200 1 0
for i in 1 2 3; do
  t0=$(date +%s.%N)
  PENGK_TIMING=1 peng-motif_amd/host/peng_motif $D/s.fa -w $W -o $D/o.meme -j $D/o.json > $D/out.txt 2> $D/err.txt
  t1=$(date +%s.%N)
  python3 -c "print(\"process wall %.3f s\" % ($t1 - $t0))"; grep -E "timing" $D/err.txt | tr '\n' ';'; echo
done
md5sum $D/o.meme $D/out.txt
PENGK_NO_STREAMING=1 PENGK_TIMING=1 peng-motif_amd/host/peng_motif $D/s.fa -w $W -o $D/o2.meme > $D/out2.txt 2> $D/err2.txt; grep -E "timing" $D/err2.txt | tr '\n' ';'; echo; md5sum $D/o2.meme $D/out2.txt
rm -rf $D
