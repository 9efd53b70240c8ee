#!/bin/bash
# sweep of the serial EM's table budget (MiB of weight tables in flight) and stream count on the 1000-PWM stress (GPU box)
for lanes in 2 3; do
for mb in 64 96 128 192 256 384; do
  python bench.py --em-overlap $lanes --em-table-budget-mb $mb --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --k4-patterns 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['components']; print('streams $lanes budget_mb $mb em_ms', c['em_ms'], 'stress serial ms', c['em_stress_serial_mode_ms'])"
done; done
