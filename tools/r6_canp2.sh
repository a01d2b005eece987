mkdir -p gpurun_out/r6i
export DMX_EXPERIMENTS=1
run() { timeout 300 env DMX_LIB=$PWD/demuxlet_amd/$1 $2 python bench.py --config 2 --only --no-cpu-baseline --steps 20 --warmup 5 >/dev/null 2>gpurun_out/r6i/err.txt; echo "$1 $2 $(python tools/bench_brief.py | cut -c1-70)" >> gpurun_out/r6i/canp2.txt; }
run libdmx_ct3.so DMX_K1_NO_CANP=1
run libdmx_ct3.so DMX_K1_CANP_MINW6=1
run libdmx_ct3.so "DMX_K1_NO_CANP=1 DMX_K1_CW=1"
