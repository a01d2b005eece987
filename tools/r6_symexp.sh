mkdir -p gpurun_out/r6b
tools/micro/write_rate /tmp 1100 8 > gpurun_out/r6b/write_rate.txt 2>&1
ls /dev/shm > /dev/null 2>&1 && tools/micro/write_rate /dev/shm 1100 8 > gpurun_out/r6b/write_rate_shm.txt 2>&1
export DMX_EXPERIMENTS=1
run() { # name lib extra-env cfg
  env DMX_LIB=$2 $3 python bench.py --config $4 --fast --only --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2> /dev/null; echo "$1 cfg$4: $(python tools/bench_brief.py)" >> gpurun_out/r6b/sym.txt
}
for c in 3 5; do
run nopipe $PWD/demuxlet_amd/libdmx.so DMX_SYM_NO_PIPE=1 $c
run nb3 $PWD/demuxlet_amd/libdmx.so DMX_X=1 $c
run nb2 $PWD/demuxlet_amd/libdmx_nb2.so DMX_X=1 $c
run nb5 $PWD/demuxlet_amd/libdmx_nb5.so DMX_X=1 $c
run nb9 $PWD/demuxlet_amd/libdmx_nb9.so DMX_X=1 $c
done
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "fast_mode" 2>&1 | tail -5 > gpurun_out/r6b/tests_fast.log
python -m pytest tests/test_bench_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r6b/tests_bench.log
