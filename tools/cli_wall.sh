#!/bin/bash
# GPU box: wall-clock of the whole `demuxlet` run (BAM + VCF in, four files out) on tools/make_cli_bench.py's job, with and without the
# HIP context warm-up that runs beside the scan.
D=${TMPDIR:-/tmp}/dmx_scan_bench; mkdir -p $D
[ -f $D/bench.bam ] || python tools/make_cli_bench.py $D 2000000 60000 16 3000
for mode in "DMX_NOP=1" "DMX_NO_WARM_UP=1" "DMX_NOP=2" "DMX_NO_WARM_UP=1"; do
  s=$(date +%s.%N)
  env $mode DMX_CLI_TIMING=1 DMX_E2E_TIMING=1 demuxlet_amd/demuxlet --sam $D/bench.bam --vcf $D/bench.vcf --field GT --out $D/o_w 2> $D/err.txt
  e=$(date +%s.%N)
  echo "$mode: wall $(python3 -c "print(round($e - $s, 3))") s; $(grep -o 'scan timing ([0-9]* threads, windowed): total [0-9.]* s' $D/err.txt); $(grep -o '{"dmx_demuxlet_run".*' $D/err.txt | cut -c1-200)"
done
md5sum $D/o_w.best
