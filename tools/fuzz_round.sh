#!/bin/bash
# GPU box: the round's fuzz / soak pass on the final build (summaries only; a failing tool stops with its case on stderr).
#   tools/fuzz_round.sh <seed base>     -> gpurun_out/fuzz_round.txt
S=${1:-9900}; OUT=gpurun_out/fuzz_round.txt; : > $OUT
run() { echo "== $*" >> $OUT; ( "$@" 2>&1 | tail -3 ) >> $OUT; }
run python tools/fuzz_parity.py 500 $((S+1))
DMX_FUZZ_FAST=1 run python tools/fuzz_parity.py 300 $((S+2))
run python tools/fuzz_e2e.py 200 $((S+3))
DMX_FUZZ_FAST=1 run python tools/fuzz_e2e.py 200 $((S+4))
run python tools/fuzz_ties.py 600 $((S+5))
DMX_FUZZ_FAST=1 run python tools/fuzz_ties.py 300 $((S+6))
run python tools/soak_determinism.py 8
cat $OUT
