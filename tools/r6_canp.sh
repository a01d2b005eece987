# GPU box: k_singlet_canp (producers / consumer K1) — parity with k_singlet_can, then cfg2 with and without it, 4 / 6 / 8 wavefronts per SIMD, timing builds
mkdir -p gpurun_out/r6i
export DMX_EXPERIMENTS=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "producer_consumer_k1" 2>&1 | tail -15 > gpurun_out/r6i/tests_canp.log
run() { timeout 300 env DMX_LIB=$PWD/demuxlet_amd/$1 $2 python bench.py --config 2 --only --no-cpu-baseline --steps 20 --warmup 5 >/dev/null 2>gpurun_out/r6i/err.txt; echo "$1 $2 $(python tools/bench_brief.py | cut -c1-70)" >> gpurun_out/r6i/canp.txt; }
run libdmx.so DMX_X=1
run libdmx.so DMX_K1_NO_CANP=1
run libdmx.so DMX_K1_CANP_MINW6=1
run libdmx.so DMX_K1_CANP_MINW8=1
run libdmx_ct1.so DMX_K1_CANP_MINW6=1
run libdmx_ct3.so DMX_K1_CANP_MINW6=1
run libdmx_ct3.so DMX_K1_CANP_MINW8=1
