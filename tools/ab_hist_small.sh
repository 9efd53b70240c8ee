for r in 1 2; do
for lib in ablation_libs/head.so peng-motif_amd/libpengk.so; do
for n in 1250000 10000000; do
PENGK_LIB=$PWD/$lib timeout -k 10 200 python bench.py --nseq $n --pwms 2 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --strong 0 --config3-steps 0 --pipelined 0 --em-stress-pwms 0 > gpurun_out/ab.json 2> gpurun_out/ab.err || exit 1
python -c "
import json
d=json.load(open('gpurun_out/ab.json')); c=d['components']
print('$lib', $n, 'step', d['ms_per_step'], 'count_ms', c['count_ms'], d['checks_ok']['ok'])"
done; done; done
PENGK_LIB=$PWD/ablation_libs/head.so timeout -k 10 200 python bench.py --W 12 --nseq 1562500 --pwms 2 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --strong 0 --config3-steps 0 --pipelined 0 --em-stress-pwms 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('head W12 1.56M', d['ms_per_step'], d['components']['count_ms'])"
timeout -k 10 200 python bench.py --W 12 --nseq 1562500 --pwms 2 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --strong 0 --config3-steps 0 --pipelined 0 --em-stress-pwms 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('new  W12 1.56M', d['ms_per_step'], d['components']['count_ms'])"
