#!/bin/bash
# wall time of peng_motif at a given W on MafK.fasta (GPU box): tools/cli_timing_w.sh 12 [extra args]
W=$1; shift
( time PENGK_TIMING=1 timeout -k 5 400 peng-motif_amd/host/peng_motif tests/golden/MafK.fasta -w $W -o /tmp/o.meme "$@" > /tmp/o.stdout ) 2>&1 | grep -E "timing|real|rror"
grep -c MOTIF /tmp/o.meme
