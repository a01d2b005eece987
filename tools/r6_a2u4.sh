# GPU box: k_doublet_diag (two barcodes per wavefront, both alphas per lane) against the one-wavefront-per-barcode diagonal (DMX_A2U_DIAG_WAVE=1)
mkdir -p gpurun_out/r6m
export DMX_EXPERIMENTS=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "symmetric_strict or unordered_pair or cfg3 or phase1_final_tables_leave_the_strict" 2>&1 | tail -4 > gpurun_out/r6m/tests_a2u4.log
for v in "DMX_X=1" "DMX_A2U_DIAG_WAVE=1" "DMX_X=1" "DMX_A2U_DIAG_WAVE=1"; do
  timeout 600 env $v python bench.py --config 3 --only --no-cpu-baseline --steps 5 --warmup 1 >/dev/null 2>gpurun_out/r6m/err.txt
  echo "$v $(python tools/bench_brief.py | cut -c1-170)" >> gpurun_out/r6m/a2u_diag2.txt
done
