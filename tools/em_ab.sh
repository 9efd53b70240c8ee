#!/bin/bash
# usage (GPU box): tools/em_ab.sh ROUNDS "ENV_A" "ENV_B" ... -> tools/em_probe.py (16 PWMs, 1000 PWMs, W = 12) under each environment
# in turn (e.g. "PENGK_LIB=$PWD/ablation_libs/x.so"), medians per environment and case
R=$1; shift
for r in $(seq 1 $R); do
  for envs in "$@"; do
    a=$(env $envs python tools/em_probe.py --reps 30 | head -1 | sed 's/.*median \([0-9.]*\) ms.*/\1/')
    b=$(env $envs python tools/em_probe.py --pwms 1000 --reps 3 | head -1 | sed 's/.*median \([0-9.]*\) ms.*/\1/')
    c=$(env $envs python tools/em_probe.py --W 12 --reps 5 | head -1 | sed 's/.*median \([0-9.]*\) ms.*/\1/')
    echo "round $r [${envs}] 16 PWMs $a ms   1000 PWMs $b ms   W=12 $c ms"
  done
done
