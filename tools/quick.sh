#!/bin/bash
# usage (GPU box): tools/quick.sh <tag> [pytest -k expr]  -> count parity subset, then rocprofv3 kernel stats of a short bench
T=$1; K=${2:-count}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "$K" > gpurun_out/${T}_t.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/${T}_t.log
HEADN=${HEADN:-7} tools/kstats.sh ${T}_ks
