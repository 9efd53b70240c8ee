#!/bin/bash
# A/B of the serial EM's summation (option em_serial_scan: 2 = blocks evaluated ahead of their chain, 1 = block after
# block) through bench.py: the step's 16 PWMs, the 1000-PWM stress, and with W12=1 the 16 PWMs of a configs[3] shard.
# usage (GPU box): tools/ab_scanmode.sh <rounds> [modes...]   (2o3 = mode 2 with em_overlap = 3 streams)
R=$1; shift; M=${@:-2 1}
for r in $(seq 1 $R); do
  for m in $M; do
    python bench.py --em-serial-scan ${m%o*} --em-overlap $([ "${m#*o}" = "$m" ] && echo 0 || echo ${m#*o}) --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --k4-patterns 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['components']; print('round $r scan=$m W=10 em_ms(16 PWMs x 10 it)', c['em_ms'], 'stress serial ms', c['em_stress_serial_mode_ms'], 'stress fast ms', c['em_stress_ms'])"
    if [ -n "$W12" ]; then
    python bench.py --em-serial-scan ${m%o*} --em-overlap $([ "${m#*o}" = "$m" ] && echo 0 || echo ${m#*o}) --W 12 --nseq 12500000 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --k4-patterns 0 --em-stress-pwms 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['components']; print('round $r scan=$m W=12 em_ms(16 PWMs x 10 it)', c['em_ms'], 'count_ms', c['count_ms'])"
    fi
  done
done
