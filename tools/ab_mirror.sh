D=/dev/shm/pengk_ab_$$; mkdir -p $D
tools/synth_fasta $D/s.fa 12500000 200 1 0
for mb in 16 100000 16 100000; do
  echo "== pinned max $mb MB"
  PENGK_MIRROR_PINNED_MAX_MB=$mb PENGK_TIMING=1 peng-motif_amd/host/peng_motif $D/s.fa -w 12 -o $D/o$mb.meme > $D/out$mb.txt 2> $D/err.txt
  grep -E "seed|host|ranking|walk|total|base patterns|hill" $D/err.txt
done
md5sum $D/*.meme $D/out*.txt
rm -rf $D
