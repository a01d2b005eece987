# GPU box: k_singlet_canp (producers / consumer K1; experiment) — parity with k_singlet_can, then cfg2 with and without it, with and without the staged streams
mkdir -p gpurun_out/r6i
export DMX_EXPERIMENTS=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "producer_consumer_k1" 2>&1 | tail -15 > gpurun_out/r6i/tests_canp.log
run() { timeout 300 env $1 python bench.py --config 2 --only --no-cpu-baseline --steps 20 --warmup 5 >/dev/null 2>gpurun_out/r6i/err.txt; echo "$1 $(python tools/bench_brief.py | cut -c1-70)" >> gpurun_out/r6i/canp4.txt; }
run "DMX_K1_CANP=1"
run "DMX_K1_CANP=1 DMX_K1_CANP_NO_STAGE=1"
run "DMX_K1_CANP=1 DMX_K1_CANP_MINW4=1"
run "DMX_X=1"
run "DMX_K1_CANP=1"
