#!/bin/bash
export DMX_EXPERIMENTS=1   # the kernel-variant switches are honoured only with this (dmx_engine_create)
# K2 throughput against panel width (run on the GPU box): tools/sweep_v.sh "64 96 128" "GP GT" 400
for v in $1; do for fld in $2; do
  python bench.py --config 3 --cells $3 --samples $v --field $fld --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('V', $v, '$fld', d['fp64_valu']['kernel_ms'], '%.3e pair-evals/s' % d['pair_evals_per_s'])"
done; done
