#!/bin/bash
# what kind of box is this?  partition modes, clocks, power cap -- beside the three EM timings that vary from box to box
rocm-smi --showmemorypartition --showcomputepartition 2>&1 | grep -v "^=\|^$" | head -6
rocm-smi --showperflevel --showmaxpower --showclocks 2>&1 | grep -E "GPU\[0\]" | head -12
cat /sys/class/drm/card*/device/current_memory_partition 2>/dev/null | head -2
timeout -k 5 100 python tools/em_probe.py --scan 2 --W 10 --pwms 16 --reps 30 | grep -v sha | cut -c1-110
timeout -k 5 100 python tools/em_probe.py --scan 2 --W 12 --pwms 16 --reps 5 | grep -v sha | cut -c1-110
rocm-smi --showclocks 2>&1 | grep -E "GPU\[0\]" | head -8
