#!/bin/bash
# W = 14 end to end against the compiled reference (GPU box): the reference needs ~10 GB and 5-9 minutes per run at this pattern
# length, so this is a soak, not a suite test.  usage: tests/tools/cli_w14_differential.sh [extra flags]
T=$(mktemp -d); R=$PWD/oracle/_ref/peng_motif_ref; H=$PWD/peng-motif_amd/host/peng_motif; F=$PWD/tests/golden/MafK_100seqs.fasta
s=$(date +%s)
"$R" $F -w 14 "$@" -o $T/r.meme -j $T/r.json > $T/r.out 2> $T/r.err &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 30; echo "... reference running, $(( $(date +%s) - s )) s"; done
wait $pid; rr=$?
t_ref=$(( $(date +%s) - s )); s=$(date +%s)
"$H" $F -w 14 "$@" -o $T/h.meme -j $T/h.json > $T/h.out 2> $T/h.err; hr=$?
t_here=$(( $(date +%s) - s ))
so=same; cmp -s $T/r.out $T/h.out || so=DIFF
se=same; cmp -s $T/r.err $T/h.err || se=differs
sm=same; cmp -s $T/r.meme $T/h.meme || sm=DIFF
sj=same; cmp -s $T/r.json $T/h.json || sj=DIFF
echo "W=14 MafK_100seqs $*: rc ref=$rr here=$hr stdout=$so stderr=$se meme=$sm json=$sj motifs=$(grep -c MOTIF $T/r.meme) reference ${t_ref} s, here ${t_here} s"
rm -rf $T
