#!/bin/bash
# GPU box (or any host): throughput of the `demuxlet` binary's BAM x VCF scan (rows f1-f3) on tools/make_cli_bench.py's job,
# one thread / read by read against the windowed scan on all cores; the two --pileup-only dumps must be the same bytes.
#   tools/scan_bench.sh [reads] [snps] [samples] [barcodes]
set -e
D=${TMPDIR:-/tmp}/dmx_scan_bench; mkdir -p $D
R=${1:-2000000}; S=${2:-60000}; V=${3:-16}; B=${4:-3000}
[ -f $D/bench.bam ] || python tools/make_cli_bench.py $D $R $S $V $B
nproc
for mode in "DMX_THREADS=1" "DMX_SCAN_SEQUENTIAL=1" "-" "DMX_THREADS=8" "DMX_THREADS=32"; do
  [ "$mode" = "-" ] && mode="DMX_NOP=1"
  n=$(echo $mode | tr -c 'A-Za-z0-9' '_')
  for rep in 1 2; do
    env $mode DMX_CLI_TIMING=1 demuxlet_amd/demuxlet --sam $D/bench.bam --vcf $D/bench.vcf --field GT --out $D/o_$n --pileup-only 2>&1 | grep "scan timing" | sed "s/^.*scan timing/$mode: scan timing/"
  done
  md5sum $D/o_$n.pileup.txt | cut -c1-32
done
