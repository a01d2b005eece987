#!/bin/bash
OUT=gpurun_out/fuzz_long.txt; : > $OUT
run() { echo "== $*" >> $OUT; ( "$@" 2>&1 | tail -2 ) >> $OUT; }
run python tools/fuzz_ties.py 2000 9911
DMX_FUZZ_FAST=1 run python tools/fuzz_ties.py 1000 9912
run python tools/fuzz_parity.py 1000 9913
DMX_FUZZ_FAST=1 run python tools/fuzz_parity.py 800 9914
DMX_FUZZ_DEEP=1 run python tools/fuzz_parity.py 200 9915
DMX_FUZZ_CLSP=1 run python tools/fuzz_parity.py 200 9916
DMX_FUZZ_CLSP=1 DMX_FUZZ_FAST=1 run python tools/fuzz_parity.py 200 9917
run python tools/fuzz_e2e.py 400 9918
DMX_FUZZ_FAST=1 run python tools/fuzz_e2e.py 400 9919
cat $OUT
