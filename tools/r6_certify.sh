# GPU box: k_certify with the final-value tables (round 6) against the seeds-only form (DMX_CERTIFY_NO_FINALS=1), and k_doublet_sym's timing ablations
mkdir -p gpurun_out/r6d
export DMX_EXPERIMENTS=1
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "certify" 2>&1 | tail -5 > gpurun_out/r6d/tests_certify.log
python -m pytest tests/test_gpu_ties.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r6d/tests_ties.log
for spec in "3 0 --fast" "5 0 --fast" "4 12500 "; do set -- $spec
  for v in "" 1; do
    if [ -n "$v" ]; then export DMX_CERTIFY_NO_FINALS=1; else unset DMX_CERTIFY_NO_FINALS; fi
    c=""; [ "$2" != "0" ] && c="--cells $2"
    python bench.py --config $1 $c $3 --only --no-cpu-baseline --steps 5 --warmup 2 >/dev/null 2>gpurun_out/r6d/err.txt
    echo "nofinals=$v $(python tools/bench_brief.py)" >> gpurun_out/r6d/certify.txt
  done
done
unset DMX_CERTIFY_NO_FINALS
for a in NONE P2 U RD P1 00; do
  env DMX_LIB=$PWD/demuxlet_amd/libdmx_abl.so DMX_SYM_ABLATE_$a=1 python bench.py --config 3 --fast --only --no-cpu-baseline --steps 5 --warmup 2 >/dev/null 2>gpurun_out/r6d/err.txt
  echo "ablate=$a $(python tools/bench_brief.py)" >> gpurun_out/r6d/ablate.txt
done
