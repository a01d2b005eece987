# GPU box: phase-1 final-value table, all-or-nothing per tile: tests, then cfg3 / cfg5 in both modes with and without it
mkdir -p gpurun_out/r6h
export DMX_EXPERIMENTS=1
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "phase1_final or certify_seed" 2>&1 | tail -8 > gpurun_out/r6h/tests_finals.log
for spec in "3 " "5 " "3 --fast" "5 --fast"; do set -- $spec
  for v in "" 1; do
    if [ -n "$v" ]; then export DMX_A2_NO_FINALS=1 DMX_SYM_NO_FINALS=1 DMX_CERTIFY_NO_FINALS=1; else unset DMX_A2_NO_FINALS DMX_SYM_NO_FINALS DMX_CERTIFY_NO_FINALS; fi
    python bench.py --config $1 $2 --only --no-cpu-baseline --steps 5 --warmup 2 >/dev/null 2>gpurun_out/r6h/err.txt
    echo "nofinals=$v $(python tools/bench_brief.py | cut -c1-220)" >> gpurun_out/r6h/finals.txt
  done
done
