# GPU box: k_doublet_a2u16 + k_doublet_diag<5,16> (cfg5 STRICT) — parity with k_doublet_a2, then cfg5 STRICT with and without
mkdir -p gpurun_out/r6n
export DMX_EXPERIMENTS=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "unordered_pair or symmetric_strict or cfg5 or phase1_final_tables_leave_the_strict" 2>&1 | tail -6 > gpurun_out/r6n/tests.log
for v in "DMX_X=1" "DMX_A2_NO_SYMU=1" "DMX_X=1" "DMX_A2_NO_SYMU=1"; do
  timeout 600 env $v python bench.py --config 5 --only --no-cpu-baseline --steps 6 --warmup 2 >/dev/null 2>gpurun_out/r6n/err.txt
  echo "$v $(python tools/bench_brief.py | cut -c1-300)" >> gpurun_out/r6n/a2u16.txt
done
