#!/bin/bash
# GPU box: a run with hundreds of motifs (MafK.fasta, z threshold 4, up to 1000 optimised patterns) through the compiled
# reference and through peng_motif -- the device similarity grid picks the merge candidates here -- compared byte for byte.
cd $GRAFT_REPO_ROOT
F="tests/golden/MafK.fasta -w 10 -t 4 --max-optimized-patterns 1000 --minimum-processed-patterns 1000"
( time oracle/_ref/peng_motif_ref $F -o /tmp/ref.meme -j /tmp/ref.json > /tmp/ref.out 2>/tmp/ref.err ) 2>&1 | grep real
( time peng-motif_amd/host/peng_motif $F -o /tmp/our.meme -j /tmp/our.json > /tmp/our.out 2>/tmp/our.err ) 2>&1 | grep real
echo "em lines: $(grep -c '^em:' /tmp/ref.out) / $(grep -c '^em:' /tmp/our.out); merges: $(grep -c '^merge:' /tmp/ref.out) / $(grep -c '^merge:' /tmp/our.out)"
cmp /tmp/ref.out /tmp/our.out && cmp /tmp/ref.meme /tmp/our.meme && cmp /tmp/ref.json /tmp/our.json && echo "IDENTICAL stdout, MEME, JSON"
