mkdir -p gpurun_out/r6i
export DMX_EXPERIMENTS=1
run() { timeout 300 env DMX_LIB=$PWD/demuxlet_amd/$1 $2 python bench.py --config 2 --only --no-cpu-baseline --steps 20 --warmup 5 >/dev/null 2>gpurun_out/r6i/err.txt; echo "$1 $2 $(python tools/bench_brief.py | cut -c1-70)" >> gpurun_out/r6i/canp3.txt; }
for l in pa4 pa8 pa16; do
run libdmx_$l.so DMX_K1_CANP_MINW6=1
run libdmx_$l.so DMX_X=1
done
run libdmx_pa8.so DMX_K1_NO_CANP=1
timeout 600 env DMX_LIB=$PWD/demuxlet_amd/libdmx_pa8.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "producer_consumer_k1" 2>&1 | tail -3 > gpurun_out/r6i/tests_canp_pa8.log
