# GPU box: phase 1's read loop from the seed table in k_doublet_sym / k_doublet_a2 (five-value form) — tests, then cfg5 / cfg3 in both modes with and without
mkdir -p gpurun_out/r6l
export DMX_EXPERIMENTS=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "phase1_final or certify_seed or symmetric_strict" 2>&1 | tail -8 > gpurun_out/r6l/tests_seeds.log
for spec in "5 " "5 --fast" "3 " "3 --fast" "6 --fast"; do set -- $spec
  for v in "DMX_X=1" "DMX_A2_NO_SEEDS=1 DMX_SYM_NO_SEEDS=1" "DMX_X=1"; do
    timeout 600 env $v python bench.py --config $1 $2 --only --no-cpu-baseline --steps 5 --warmup 2 >/dev/null 2>gpurun_out/r6l/err.txt
    echo "$v $(python tools/bench_brief.py | cut -c1-230)" >> gpurun_out/r6l/seeds.txt
  done
done
