# GPU box: timing-only builds of k_doublet_sym (results WRONG): one FMA fewer in the polynomial (1), k ln 2 by LDS look-up + add (2), both (3)
mkdir -p gpurun_out/r6f
export DMX_EXPERIMENTS=1
run() { env DMX_LIB=$2 python bench.py --config $3 --fast --only --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2> gpurun_out/r6f/err.txt; echo "$1 $(python tools/bench_brief.py | cut -c1-150)" >> gpurun_out/r6f/timing.txt; }
for c in 3 5; do
run base $PWD/demuxlet_amd/libdmx.so $c
run tim1 $PWD/demuxlet_amd/libdmx_tim1.so $c
run tim2 $PWD/demuxlet_amd/libdmx_tim2.so $c
run tim3 $PWD/demuxlet_amd/libdmx_tim3.so $c
done
