#!/bin/bash
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cli_prof -- $R/peng-motif_amd/host/peng_motif $R/tests/golden/MafK.fasta -w 10 -o /tmp/o.meme > /tmp/o.stdout 2> /tmp/o.err
cd $R
cat gpurun_out/cli_prof/*/*kernel_stats.csv | sed "s/(anonymous namespace):://; s/(unsigned[^\"]*\"/\"/; s/(int[^\"]*\"/\"/; s/(float[^\"]*\"/\"/" | cut -c1-130 | head -12
