#!/bin/bash
# what is left at the end of peng_motif on the bench's FASTA: resident memory, and the pieces of an orderly teardown (GPU box)
D=/dev/shm/pengk_td_$$; mkdir -p $D
tools/synth_fasta $D/s.fa 10000000 200 1 0
for i in 1 2; do
PENGK_FULL_TEARDOWN=$((i-1)) PENGK_TIMING=1 peng-motif_amd/host/peng_motif $D/s.fa -w 10 -o $D/o.meme 2>&1 >/dev/null | grep -E "released|destroyed|cleanup|total|resident" | tr "\n" ";"; echo
done
rm -rf $D
