R=$PWD; D=$R/gpurun_out/emraw_$$
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/tools/em_probe.py --reps 3 "$@" > $D.log 2>&1
cd $R; python tools/em_rawtrace.py $D 70; rm -rf $D $D.log
