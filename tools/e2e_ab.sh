#!/bin/bash
# usage (GPU box): tools/e2e_ab.sh ROUNDS "ENV_A" "ENV_B" ["ENV_C" ...] -> peng_motif on the bench's FASTA, the
# configurations taking turns run by run (the box's noise hits all of them alike), wall times and medians per configuration
R=$1; shift
D=/dev/shm/pengk_e2e_$$; mkdir -p $D
tools/synth_fasta $D/s.fa 10000000 200 1 0
peng-motif_amd/host/peng_motif $D/s.fa -w 10 -o $D/o.meme > /dev/null 2>&1   # warm: page cache, code objects
declare -A T
for r in $(seq 1 $R); do
  i=0
  for envs in "$@"; do
    t0=$(date +%s.%N)
    env $envs PENGK_TIMING=1 peng-motif_amd/host/peng_motif $D/s.fa -w 10 -o $D/o.meme -j $D/o.json > /dev/null 2> $D/err_$i.txt
    t1=$(date +%s.%N)
    T[$i]="${T[$i]} $(python3 -c "print('%.3f' % ($t1 - $t0))")"
    tot=$(grep "timing. total" $D/err_$i.txt | tr -s ' ' | cut -d' ' -f3)
    T2[$i]="${T2[$i]} $tot"
    i=$((i+1))
  done
done
i=0
for envs in "$@"; do
  echo "[${envs:-default}] wall:${T[$i]}"
  python3 -c "import sys,statistics as s; v=[float(x) for x in sys.argv[1:]]; print('    median %.3f  min %.3f  max %.3f' % (s.median(v), min(v), max(v)))" ${T[$i]}
  echo "    main() total:${T2[$i]}"
  i=$((i+1))
done
rm -rf $D
