#!/bin/bash
# usage (GPU box): tools/em_stress_kstats.sh <tag> <lib.so>  -> rocprofv3 kernel stats of the bench incl. the 1000-PWM serial EM stress
export TMPDIR=/tmp
R=$PWD; O=$1; LIB=$2
cd /tmp && PENGK_LIB=$R/$LIB timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$O -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --k4-patterns 0 > $R/gpurun_out/$O.log 2>&1
echo "rc=$?"; cd $R
cat gpurun_out/$O/*/*kernel_stats.csv | grep -E "em_|Name" | sed "s/(anonymous namespace):://; s/(unsigned[^\"]*\"/\"/; s/(float[^\"]*\"/\"/; s/(int[^\"]*\"/\"/" | cut -c1-150
