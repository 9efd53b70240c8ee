#!/bin/bash
# A/B of library variants on the EM stress (BASELINE configs[4]) in one GPU session: tools/ab_em.sh <rounds> a.so b.so ...
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    PENGK_LIB=$PWD/$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --k4-patterns 0 --em-fast 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['components']; print('round $r $v em_stress_ms', c['em_stress_ms'], 'em_ms(16 PWMs)', c['em_ms'], 'serial stress', c['em_stress_serial_mode_ms'])"
  done
done
