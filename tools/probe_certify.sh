#!/bin/bash
export DMX_EXPERIMENTS=1   # the kernel-variant switches are honoured only with this (dmx_engine_create)
# GPU box: K3b (k_certify) time at cfg3 / cfg5 / cfg4-shard sizes, FAST, for the variants selected by environment switches.
for cfg in 3 5 4; do
  cells=""; [ $cfg = 4 ] && cells="--cells 12500"
  for v in "" "DMX_CERTIFY_MINW3=1" "DMX_CERTIFY_NO_GT=1"; do
    env $v python bench.py --config $cfg $cells --fast --only --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg$cfg', '$v', d['ms_per_step'], d['fp64_valu']['kernel_ms'])"
  done
done
