#!/bin/bash
export DMX_EXPERIMENTS=1   # the kernel-variant switches are honoured only with this (dmx_engine_create)
# GPU box: per-kernel times of bench configurations for several library variants / environment switches.
#   tools/probe_variants.sh "<cfg> [bench flags]" "<VAR=val ...>" ...      ("-" = no switch)
spec=$1; shift
for v in "$@"; do
  [ "$v" = "-" ] && v=""
  env $v python bench.py --config $spec --only --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['fp64_valu']['kernel_ms']; print('cfg $spec | $v |', round(d['ms_per_step'],2), {a: round(b,2) for a,b in k.items() if isinstance(b,(int,float)) and not isinstance(b,bool) and not a.startswith('torch')})"
done
