mkdir -p gpurun_out/r6j
export DMX_EXPERIMENTS=1
run() { timeout 600 env DMX_LIB=$PWD/demuxlet_amd/$1 $2 python bench.py --config 3 --only --no-cpu-baseline --steps 4 --warmup 1 >/dev/null 2>gpurun_out/r6j/err.txt; echo "$1 $2 $(python tools/bench_brief.py | cut -c1-150)" >> gpurun_out/r6j/a2s3.txt; }
run libdmx.so "DMX_A2_SYM=1 DMX_A2S_SUB8=1"
run libdmx.so "DMX_A2_SYM=1 DMX_A2S_SUB8=1 DMX_K1_AFTER_K2=1"
run libdmx_pu2.so "DMX_A2_SYM=1 DMX_A2S_SUB8=1"
run libdmx_pu2.so "DMX_A2_SYM=1"
run libdmx.so "DMX_X=1"
