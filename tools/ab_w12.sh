#!/bin/bash
# A/B of library variants on the count of a configs[3] shard (12.5M x 200 bp, W = 12): tools/ab_w12.sh <rounds> a.so b.so ...
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    PENGK_LIB=$PWD/$v python bench.py --W 12 --nseq 12500000 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --k4-patterns 0 --em-stress-pwms 0 --pwms 2 --em-iters 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['components']; print('round $r $v W=12 count_ms', c['count_ms'])"
  done
done
