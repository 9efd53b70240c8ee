#!/bin/bash
# A/B timing of library variants on ONE GPU box, interleaved rounds (cdna_hip_programming.md rule 24).
# usage: tools/ab.sh <rounds> <variant.so> [<variant.so> ...]   (variants built with PENGK_BUILD_OUT=... build.py --force)
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    ms=$(PENGK_LIB=$PWD/$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --em-stress-pwms 0 --k4-patterns 0 ${AB_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['components']['count_ms'])")
    echo "round $r $v count_ms $ms"
  done
done
