#!/bin/bash
# usage (GPU box): tools/profile_round.sh <tag> [bench size args]   e.g. r03_final      |  r03_w12 --W 12 --nseq 12500000
# rocprofv3 evidence of one round: kernel stats of the bench at that size, then separate --pmc passes (HBM traffic, SQ
# issue counters) of a one-step run -- counters are collected in their own runs, with --kernel-trace only.
# Everything lands in gpurun_out/<tag>/; tools/profile_summary.py condenses it into profiles/ (one entry per
# configuration in profiles/traffic_by_config.json).
set -e
export TMPDIR=/tmp
R=$PWD; T=$1; shift; CFG="$@"; OUT=$R/gpurun_out/$T
mkdir -p $OUT
BARGS="--no-cpu-baseline --no-e2e --k4-patterns 0 --config3-steps 0 $CFG"   # (the configs[3] leg has its own tag: r0N_w12)
STRESS=""; [ -n "$CFG" ] && STRESS="--em-stress-pwms 0"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 3 --warmup 1 $BARGS $STRESS > $OUT/stats.log 2>&1
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 1 --warmup 1 $BARGS --em-stress-pwms 0 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
cd $R
if [ -z "$CFG" ]; then python3 bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
else python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e $CFG > $OUT/bench.json 2> $OUT/bench.err; fi
python3 tools/profile_summary.py $OUT $PENGK_COMMIT   # (no .git on the GPU box: the caller passes the commit)
