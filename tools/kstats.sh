#!/bin/bash
# usage: tools/kstats.sh <outname> [bench args...]  -> rocprofv3 kernel stats of one bench run (GPU box)
export TMPDIR=/tmp
R=$PWD; O=$1; shift
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$O -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e "$@" > $R/gpurun_out/$O.log 2>&1
echo "rc=$?"; cd $R
grep "^{" gpurun_out/$O.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['components'])"
cat gpurun_out/$O/*/*kernel_stats.csv | sed "s/(anonymous namespace):://; s/(unsigned[^\"]*\"/\"/; s/(float[^\"]*\"/\"/" | cut -c1-120 | head -${HEADN:-6}
