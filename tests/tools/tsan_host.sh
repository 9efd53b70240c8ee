#!/bin/bash
# ThreadSanitizer over the threaded host code of the path (build container; CPU build only -- GPU sanitizers are not
# available on the pool): the streaming ingest (reader threads -> pengk_pack_append -> uploader thread, madvise of packed
# chunks, the context starter) with forced chunkings, and the host channel with 1 / 2 / 4 ranks whose threads call the
# collectives at once.  Run from the repository root.  Every line must end in "ok" / "streamed == staged"; any
# ThreadSanitizer report goes to stderr and makes the run fail (halt_on_error).
set -e
T=$(mktemp -d)
H=peng-motif_amd/host
F="-O1 -g -fsanitize=thread -fno-omit-frame-pointer"
g++ -std=c++17 $F -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Ipeng-motif_amd/csrc -I$H \
    tests/tools/tsan_host_driver.cpp peng-motif_amd/csrc/pack.cpp -x c++ peng-motif_amd/csrc/comm.hip -x none \
    $H/device.cpp $H/Global.cpp $H/shared/Alphabet.cpp $H/shared/Sequence.cpp $H/shared/SequenceSet.cpp $H/shared/BackgroundModel.cpp \
    -o $T/drv -pthread -ldl -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib 2> $T/build.log || { cat $T/build.log; exit 1; }
export TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1"
for f in tests/golden/torture.fa tests/golden/MafK_100seqs.fasta tests/golden/MafK.fasta; do
  for cfg in "1 1" "4 4" "9 4" "37 16" "64 8"; do
    set -- $cfg
    PENGK_READ_CHUNKS=$1 PENGK_READ_THREADS=$2 $T/drv ingest $f 8
  done
done
PENGK_READ_CHUNKS=23 PENGK_READ_THREADS=12 $T/drv ingest tests/golden/MafK.fasta 12
PENGK_READ_CHUNKS=23 PENGK_READ_THREADS=12 PENGK_NO_HUGEPAGES=1 $T/drv ingest tests/golden/MafK.fasta 4
for w in 1 2 4; do
  port=$((20000 + RANDOM % 20000))
  pids=""
  for r in $(seq 0 $((w - 1))); do
    RANK=$r WORLD_SIZE=$w MASTER_ADDR=127.0.0.1 MASTER_PORT=$port PENGK_COMM_PORT=$port PENGK_COMM_TIMEOUT=60 $T/drv chan &
    pids="$pids $!"
  done
  for p in $pids; do wait $p; done
done
rm -rf $T
echo "tsan_host: all clean"
