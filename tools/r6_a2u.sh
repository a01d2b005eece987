# GPU box: k_doublet_a2u (k_doublet_a2 over unordered pairs + the diagonal behind it) — parity with k_doublet_a2, then cfg3 STRICT with and without it
mkdir -p gpurun_out/r6m
export DMX_EXPERIMENTS=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "symmetric_strict or phase1_final_tables_leave_the_strict" 2>&1 | tail -15 > gpurun_out/r6m/tests_a2u.log
for v in "DMX_X=1" "DMX_A2_NO_SYMU=1" "DMX_X=1" "DMX_A2_NO_SYMU=1"; do
  timeout 600 env $v python bench.py --config 3 --only --no-cpu-baseline --steps 4 --warmup 1 >/dev/null 2>gpurun_out/r6m/err.txt
  echo "$v $(python tools/bench_brief.py | cut -c1-330)" >> gpurun_out/r6m/a2u.txt
done
