#!/bin/bash
cd $GRAFT_REPO_ROOT
CLI=peng-motif_amd/host/peng_motif
echo "== merge probe: MafK W=10, many patterns"
for T in 10 4 2; do
PENGK_TIMING=1 $CLI tests/golden/MafK.fasta -w 10 -t $T --max-optimized-patterns 1000 --minimum-processed-patterns 1000 -o /tmp/m.meme > /tmp/m.out 2> /tmp/m.err
echo "t=$T: seeds $(grep -c 'optimization:' /tmp/m.out) em lines $(grep -c '^em:' /tmp/m.out) merges $(grep -c '^merge:' /tmp/m.out) motifs $(grep -c MOTIF /tmp/m.meme)"
grep -E "seed selection|hill-climb|filter|EM|merging|total" /tmp/m.err | tr '\n' ';'; echo
done
echo "== ranked_prefix probe: W=12"
tools/synth_fasta /tmp/s2m.fa 2000000 200 1 0
PENGK_TIMING=1 $CLI /tmp/s2m.fa -w 12 -o /tmp/m.meme > /tmp/m.out 2> /tmp/m.err
grep -E "seed selection|hill-climb|EM|merging|count|total|base patterns" /tmp/m.err | tr '\n' ';'; echo
PENGK_TIMING=1 $CLI tests/golden/MafK.fasta -w 12 -o /tmp/m.meme > /tmp/m.out 2> /tmp/m.err
grep -E "seed selection|hill-climb|EM|merging|total" /tmp/m.err | tr '\n' ';'; echo
