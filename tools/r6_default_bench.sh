# the driver's N = 1 command on this build: the line and the full record, then the SAME command under rocprofv3 --kernel-trace --stats
mkdir -p gpurun_out/r6_default
export TMPDIR=/tmp
DMX_BENCH_FULL=$PWD/gpurun_out/r6_default/full.json python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_default/line.json 2> gpurun_out/r6_default/err.log
echo "plain run rc=$?"
( cd /tmp && DMX_BENCH_FULL=$OLDPWD/gpurun_out/r6_default/full_under_rocprof.json rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_default -o kt -- python $OLDPWD/bench.py --gpus 1 --steps 20 --warmup 5 > $OLDPWD/gpurun_out/r6_default/line_under_rocprof.json 2> /dev/null )
echo "rocprof run rc=$?"
find /tmp/kt_default -name "*kernel_stats.csv" -exec cp {} gpurun_out/r6_default/kernel_stats.csv \;
f=$(find /tmp/kt_default -name "*kernel_trace.csv" | head -1)
python tools/summarize_default_profile.py $f > gpurun_out/r6_default/kernel_groups.csv
ls -la gpurun_out/r6_default
