#!/bin/bash
# usage (GPU box): tools/multirank_scale.sh [nseq] [W] [ranks]  -> the sharded multi-rank peng_motif at scale on ONE GPU (tcp rehearsal
# transport): plain run vs RANKS processes on the same synthetic FASTA; md5 of stdout / MEME / JSON, wall times, peak RSS per rank
N=${1:-12500000}; W=${2:-12}; R=${3:-2}
D=/dev/shm/pengk_mr_$$; mkdir -p $D
tools/synth_fasta $D/s.fa $N 200 1 0
CLI=peng-motif_amd/host/peng_motif
t0=$(date +%s.%N); PENGK_TIMING=1 $CLI $D/s.fa -w $W -o $D/p.meme -j $D/p.json > $D/p.out 2> $D/p.err; t1=$(date +%s.%N)
python3 -c "print('plain: wall %.3f s' % ($t1 - $t0))"; grep -E "peak resident|total" $D/p.err | tr '\n' ';'; echo
PORT=$((20000 + RANDOM % 20000))
t0=$(date +%s.%N)
for r in $(seq 0 $((R-1))); do
  RANK=$r WORLD_SIZE=$R LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT PENGK_COMM_TRANSPORT=tcp PENGK_TIMING=1 \
    $CLI $D/s.fa -w $W -o $D/r$r.meme -j $D/r$r.json > $D/r$r.out 2> $D/r$r.err &
done
wait; t1=$(date +%s.%N)
python3 -c "print('$R ranks on one GPU (tcp transport): wall %.3f s' % ($t1 - $t0))"
for r in $(seq 0 $((R-1))); do echo -n "rank $r: "; grep -E "peak resident|total" $D/r$r.err | tr '\n' ';'; echo " stdout bytes $(stat -c %s $D/r$r.out)"; done
md5sum $D/p.out $D/r0.out $D/p.meme $D/r0.meme $D/p.json $D/r0.json | sed "s#$D/##"
cmp -s $D/p.out $D/r0.out && cmp -s $D/p.meme $D/r0.meme && cmp -s $D/p.json $D/r0.json && echo "IDENTICAL: stdout, MEME, JSON of rank 0 == plain run" || echo "DIFFERENT"
rm -rf $D
