#!/bin/bash
# the parity EM with three (em_serial_scan 2) against two (3) launches per iteration: tools/em_ab_scan.sh [rounds]
R=${1:-2}
for r in $(seq 1 $R); do
  for scan in 2 3; do
    python tools/em_probe.py --scan $scan --W 10 --pwms 16 --reps 30
    python tools/em_probe.py --scan $scan --W 10 --pwms 1000 --reps 5 --nseq 10000000
    python tools/em_probe.py --scan $scan --W 12 --pwms 16 --reps 5 --nseq 12500000
  done
done
