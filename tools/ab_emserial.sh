#!/bin/bash
# A/B of library variants on the serial (bit-exact) EM of the bench step: tools/ab_emserial.sh <rounds> a.so b.so ...
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    PENGK_LIB=$PWD/$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --k4-patterns 0 --em-stress-pwms 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['components']; print('round $r $v serial em_ms(16 PWMs x 10 it)', c['em_ms'])"
  done
done
