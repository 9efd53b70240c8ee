#!/bin/bash
for rows in 0 1; do for st in 1 2; do
  python tools/em_probe.py --scan 3 --rows $rows --streams $st --W 10 --pwms 16 --reps 30 | grep -v sha
done; done
for rows in 0 1; do
  python tools/em_probe.py --scan 3 --rows $rows --W 10 --pwms 1000 --reps 5 --nseq 10000000 | grep -v sha
  python tools/em_probe.py --scan 3 --rows $rows --W 12 --pwms 16 --reps 5 --nseq 12500000 | grep -v sha
done
