#!/bin/bash
# PMC passes for the count kernels (run on the GPU box): separate rocprofv3 --pmc runs, kernel-trace only.
set -e
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/$1
shift
mkdir -p $OUT
cd /tmp
i=0
for pmc in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --nseq ${NSEQ:-2000000} > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
cd $R
python3 tools/pmc_summary.py $OUT
