#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats and PMC passes for the bench command.
# Output: gpurun_out/prof_<tag>/{kernel_stats.csv, pmc_*.csv, bench.json}; copy what should be judged into profiles/.
set -u
TAG=${1:-r01}; CFG=${2:-2}; EXTRA=${3:-}
export TMPDIR=/tmp
OUT=gpurun_out/prof_${TAG}_cfg${CFG}; mkdir -p $OUT
KRE="k_singlet|k_doublet_|k_reduce\\("
python bench.py --config $CFG $EXTRA --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$CFG -o kt -- python bench.py --config $CFG $EXTRA --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2>/dev/null
cp /tmp/kt_$CFG/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || find /tmp/kt_$CFG -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-include-regex "$KRE" --output-format csv --pmc $set -d /tmp/pmc_${CFG}_$i -o pmc -- python bench.py --config $CFG $EXTRA --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  find /tmp/pmc_${CFG}_$i -name "*counter_collection.csv" -exec cp {} $OUT/pmc_$i.csv \;
done
ls -la $OUT
