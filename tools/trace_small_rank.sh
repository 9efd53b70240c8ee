cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tr_small
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr_small -- python3 bench.py --nseq 1250000 --pwms 2 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --strong 0 --config3-steps 0 --pipelined 0 --em-stress-pwms 0 > gpurun_out/tr_small.json 2> gpurun_out/tr_small.err
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/tr_small/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
    if float(r['TotalDurationNs']) > 50000 and int(r['Calls']) >= 20:
        print('%-60s calls %4s avg %9.1f us' % (n[-60:], r['Calls'], float(r['AverageNs']) / 1e3))
PY
