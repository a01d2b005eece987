#!/bin/bash
# GPU box: per-thread stage times of the windowed scan (DMX_CLI_TIMING) and the cgroup's throttling counters around each run.
D=${TMPDIR:-/tmp}/dmx_scan_bench; mkdir -p $D
[ -f $D/bench.bam ] || python tools/make_cli_bench.py $D 2000000 60000 16 3000
nproc; cat /sys/fs/cgroup/cpu.max
for mode in "DMX_NOP=1" "DMX_THREADS=6" "DMX_THREADS=8" "DMX_THREADS=10" "DMX_THREADS=12" ${SCAN_PROBE_EXTRA}; do
  for rep in 1 2 3; do
    a=$(grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' ')
    env $mode DMX_CLI_TIMING=1 demuxlet_amd/demuxlet --sam $D/bench.bam --vcf $D/bench.vcf --field GT --out $D/o_p --pileup-only 2>&1 | grep "scan t\|process CPU" | sed "s/^.*- /$mode: /"
    b=$(grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' ')
    echo "   cpu.stat before: $a after: $b"
  done
done
