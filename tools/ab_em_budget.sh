#!/bin/bash
# serial EM stress (1000 PWMs x 10 iterations, W=10) against the weight-table budget per batch: tools/ab_em_budget.sh lib.so "96 192 384" ...
LIB=$1; shift
for b in $1; do
  PENGK_LIB=$PWD/$LIB python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --k4-patterns 0 --em-table-budget-mb $b 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['components']; print('$LIB budget_mb $b em_ms', c['em_ms'], 'stress serial ms', c['em_stress_serial_mode_ms'])"
done
