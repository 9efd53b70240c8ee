export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks12 -- python3 $R/bench.py --W 12 --nseq 12500000 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --k4-patterns 0 --em-stress-pwms 0 > $R/gpurun_out/ks12.log 2>&1
cd $R
cat gpurun_out/ks12/*/*kernel_stats.csv | sed "s/(anonymous namespace):://; s/(unsigned[^\"]*\"/\"/; s/(float[^\"]*\"/\"/; s/(int[^\"]*\"/\"/" | cut -d, -f1-4,6,7 | head -14
