# GPU box: k_doublet_sym's mix of the two FAST logs (DMX_SYM_HYB builds: 0 none, 1 every third entry through dmx_log2_lite32, 4 every second, 2 two of three, 3 all)
mkdir -p gpurun_out/r6c
export DMX_EXPERIMENTS=1
run() { # name lib cfg
  env DMX_LIB=$2 python bench.py --config $3 --fast --only --no-cpu-baseline --steps 8 --warmup 2 > /dev/null 2> gpurun_out/r6c/err_$1_$3.txt; echo "$1 cfg$3: $(python tools/bench_brief.py)" >> gpurun_out/r6c/hyb.txt
}
for rep in 1 2; do
for c in 3 5; do
run hyb0 $PWD/demuxlet_amd/libdmx_hyb0.so $c
run hyb1 $PWD/demuxlet_amd/libdmx.so $c
run hyb4 $PWD/demuxlet_amd/libdmx_hyb4.so $c
run hyb2 $PWD/demuxlet_amd/libdmx_hyb2.so $c
run hyb3 $PWD/demuxlet_amd/libdmx_hyb3.so $c
done
done
python -m pytest tests/test_dmx_log.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r6c/tests_log.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "fast_mode" 2>&1 | tail -5 > gpurun_out/r6c/tests_fast.log
