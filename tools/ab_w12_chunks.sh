#!/bin/bash
# (needs the build of profiles/r05_w12_pieces.patch) the W = 12 count of one configs[3] shard in 1 / 2 / 3 / 4 / 6 pieces (PENGK_W12_CHUNKS): count_ms, step, checks against the reference-derived row
for r in 1 2; do for c in 1 2 3 4 6; do
  PENGK_W12_CHUNKS=$c python bench.py --W 12 --nseq 12500000 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --k4-patterns 0 --config3-steps 0 --em-stress-pwms 0 --pipelined 0 --strong 0 --pwms 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chunks $c: count_ms', d['components']['count_ms'], 'step', d['ms_per_step'], 'checks_ok', d['checks_ok']['ok'], d['checks_ok'].get('parts'))"
done; done
