#!/bin/bash
# A/B of library variants on the serial (bit-exact) EM: the bench step's 16 PWMs, the 1000-PWM stress of BASELINE
# configs[4], and (W12=1) the 16 PWMs of a configs[3] shard.   tools/ab_emserial2.sh <rounds> a.so b.so ...
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    PENGK_LIB=$PWD/$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --k4-patterns 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['components']; print('round $r $v W=10 em_ms(16 PWMs x 10 it)', c['em_ms'], 'stress serial ms', c['em_stress_serial_mode_ms'], 'stress fast ms', c['em_stress_ms'])"
    if [ -n "$W12" ]; then
    PENGK_LIB=$PWD/$v python bench.py --W 12 --nseq 12500000 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --k4-patterns 0 --em-stress-pwms 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['components']; print('round $r $v W=12 em_ms(16 PWMs x 10 it)', c['em_ms'], 'count_ms', c['count_ms'])"
    fi
  done
done
