#!/bin/bash
# usage (GPU box): tools/em_trace.sh [em_probe.py arguments] -> average duration of the parity EM's kernels (rocprofv3 --kernel-trace)
R=$PWD; D=$R/gpurun_out/emtrace_$$
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/tools/em_probe.py "$@" > $D.log 2>&1
cd $R; grep "W=" $D.log; python tools/em_timeline.py $D; rm -rf $D $D.log
