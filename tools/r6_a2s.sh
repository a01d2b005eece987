# GPU box: k_doublet_a2s (symmetric STRICT, V = 32; experiment, DMX_A2_SYM=1) — parity with k_doublet_a2, then cfg3 STRICT with and without it
mkdir -p gpurun_out/r6j
export DMX_EXPERIMENTS=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "symmetric_strict" 2>&1 | tail -15 > gpurun_out/r6j/tests_a2s.log
for v in "DMX_A2_SYM=1" "DMX_X=1"; do
  timeout 600 env $v python bench.py --config 3 --only --no-cpu-baseline --steps 4 --warmup 1 >/dev/null 2>gpurun_out/r6j/err.txt
  echo "$v $(python tools/bench_brief.py | cut -c1-300)" >> gpurun_out/r6j/a2s.txt
done
