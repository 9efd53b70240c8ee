#!/bin/bash
# AddressSanitizer + UBSan over the CPU side of the path (build container; GPU ASan is not available on the pool):
# the packer (csrc/pack.cpp), the seed ranking (host/ranked_prefix.h) and the FASTA reader / background model.
# Run from the repository root.  Prints three "ok"/rc lines; any sanitizer report goes to stderr.
set -e
T=$(mktemp -d)
F="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
g++ -std=c++17 $F -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Ipeng-motif_amd/csrc tests/tools/asan_pack_driver.cpp \
    peng-motif_amd/csrc/pack.cpp -o $T/pack -pthread 2>/dev/null
$T/pack
H=peng-motif_amd/host
g++ -std=c++14 $F -Iinclude -I$H $H/tests/ranked_prefix_test.cpp -o $T/rank -pthread 2>/dev/null
$T/rank
g++ -std=c++14 $F -Iinclude -I$H $H/shared/Alphabet.cpp $H/shared/Sequence.cpp $H/shared/SequenceSet.cpp $H/shared/BackgroundModel.cpp \
    $H/tests/host_ingest_dump.cpp -o $T/ingest -pthread 2>/dev/null
for f in tests/golden/torture.fa tests/golden/MafK_100seqs.fasta tests/golden/default_sequence_set.fa tests/golden/MafK.fasta; do
  ASAN_OPTIONS=detect_leaks=0 $T/ingest $f > /dev/null && echo "ingest ok $f"
done
# the chunked reader with forced chunkings, and the sharded reader (three ranks combining their shards through files)
for c in 1 5 37; do
  PENGK_READ_CHUNKS=$c PENGK_READ_THREADS=3 ASAN_OPTIONS=detect_leaks=0 $T/ingest tests/golden/MafK.fasta > /dev/null && echo "ingest ok MafK.fasta in $c chunks"
done
# round 5: paths that are not regular files -- a pipe / /dev/stdin is spooled into a memory file, a directory is an empty set
cat tests/golden/MafK.fasta | ASAN_OPTIONS=detect_leaks=0 $T/ingest /dev/stdin > $T/pipe.out && ASAN_OPTIONS=detect_leaks=0 $T/ingest tests/golden/MafK.fasta > $T/file.out \
  && cmp -s $T/pipe.out $T/file.out && echo "ingest ok MafK.fasta through a pipe"
mkdir -p $T/dir.fa && ASAN_OPTIONS=detect_leaks=0 $T/ingest $T/dir.fa > /dev/null && echo "ingest ok a directory (empty set)"
mkdir $T/g
for r in 0 1 2; do ASAN_OPTIONS=detect_leaks=0 $T/ingest tests/golden/torture.fa $r 3 $T/g > $T/g/out$r 2> $T/g/err$r & done
wait
grep -l "views ok" $T/g/out0 $T/g/out1 $T/g/out2 | wc -l | sed "s/^/sharded ingest ok on ranks: /"
rm -rf $T
