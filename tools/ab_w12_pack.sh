# A/B of a W = 12 count change per kernel: the base library (ablation_libs/base.so) against the tree's, rocprofv3 kernel stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in ablation_libs/base.so peng-motif_amd/libpengk.so; do
export PENGK_LIB=$PWD/$lib
rm -rf gpurun_out/abp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abp -- python3 bench.py --W 12 --nseq 12500000 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --strong 0 --config3-steps 0 --pipelined 0 --em-stress-pwms 0 > gpurun_out/ab.json 2> gpurun_out/ab.err || exit 1
echo "== $lib"
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/abp/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'count_' in n or 'mirror' in n:
        print('%-60s calls %4s avg %10.1f us' % (n.split('(')[0][-60:], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
