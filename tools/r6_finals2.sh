# GPU box: the phase-1 final-value table in STRICT k_doublet_a2 (cfg3 / cfg5 STRICT) against DMX_A2_NO_FINALS=1
mkdir -p gpurun_out/r6g
export DMX_EXPERIMENTS=1
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "phase1_final" 2>&1 | tail -8 > gpurun_out/r6g/tests_finals.log
for spec in "3 0" "5 0"; do set -- $spec
  for v in "" 1 ""; do
    if [ -n "$v" ]; then export DMX_A2_NO_FINALS=1; else unset DMX_A2_NO_FINALS; fi
    python bench.py --config $1 --only --no-cpu-baseline --steps 5 --warmup 2 >/dev/null 2>gpurun_out/r6g/err.txt
    echo "nofinals=$v $(python tools/bench_brief.py | cut -c1-220)" >> gpurun_out/r6g/finals.txt
  done
done
