# GPU box: k_doublet_sym's phase 1 with the final-value table (round 6) against the read loop everywhere (DMX_SYM_NO_FINALS=1)
mkdir -p gpurun_out/r6e
export DMX_EXPERIMENTS=1
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "phase1_final or certify_seed or counted_row" 2>&1 | tail -8 > gpurun_out/r6e/tests_finals.log
for spec in "3 0 --fast" "5 0 --fast" "4 12500 --fast"; do set -- $spec
  for v in "" 1; do
    if [ -n "$v" ]; then export DMX_SYM_NO_FINALS=1; else unset DMX_SYM_NO_FINALS; fi
    c=""; [ "$2" != "0" ] && c="--cells $2"
    python bench.py --config $1 $c $3 --only --no-cpu-baseline --steps 6 --warmup 2 >/dev/null 2>gpurun_out/r6e/err.txt
    echo "nofinals=$v $(python tools/bench_brief.py)" >> gpurun_out/r6e/finals.txt
  done
done
unset DMX_SYM_NO_FINALS
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "fast_mode" 2>&1 | tail -5 > gpurun_out/r6e/tests_fast.log
