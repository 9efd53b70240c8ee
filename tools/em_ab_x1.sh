#!/bin/bash
for b0 in 1 0; do for st in 1 2; do
  python tools/em_probe.py --scan 3 --block0 $b0 --streams $st --W 10 --pwms 16 --reps 30 | grep -v sha
done; done
for b0 in 1 0; do
  python tools/em_probe.py --scan 3 --block0 $b0 --W 10 --pwms 1000 --reps 5 --nseq 10000000 | grep -v sha
  python tools/em_probe.py --scan 3 --block0 $b0 --W 12 --pwms 16 --reps 5 --nseq 12500000 | grep -v sha
done
python tools/em_probe.py --scan 2 --streams 1 --W 10 --pwms 16 --reps 30 | grep -v sha
python tools/em_probe.py --scan 2 --streams 2 --W 10 --pwms 16 --reps 30 | grep -v sha
