#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats and PMC passes for ONE bench configuration at the size given.
#   tools/profile_round.sh <tag> <cfg> ["extra bench flags"]      e.g.  tools/profile_round.sh r02 3 "--fast"
# (pass 9, round 5: the vector-L1 counters — tag look-ups, TA / TD busy — that showed what bound K1; pass 7: the LDS issue-stall and CU-busy counters VERDICT r2 asked for; a pass whose counter names this ROCm does not know leaves an empty file)
# Output: gpurun_out/prof_<tag>_cfg<N>[_fast]/{kernel_stats.csv, pmc_*.csv, bench.json}; tools/summarize_profile.py condenses
# them into profiles/.  Counters are collected in their own passes (--pmc only, no tracing), 8 SQ slots per pass; FETCH_SIZE and
# WRITE_SIZE need a pass each (MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -u
TAG=${1:-r02}; CFG=${2:-3}; EXTRA=${3:-}
export TMPDIR=/tmp
export DMX_EXPERIMENTS=1 DMX_NO_OVERLAP=1      # kernel-level profiles: K1 and K2 one after the other (bench.py otherwise runs K1 beside K2, dmx_engine_run)
SFX=""; case "$EXTRA" in *--fast*) SFX="_fast";; esac
OUT=$PWD/gpurun_out/prof_${TAG}_cfg${CFG}${SFX}; mkdir -p $OUT
KRE="k_singlet|k_doublet_|k_reduce|k_certify"
STEPS=${STEPS:-5}; PSTEPS=${PSTEPS:-2}
python bench.py --config $CFG $EXTRA --steps $STEPS --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$CFG$SFX -o kt -- python $OLDPWD/bench.py --config $CFG $EXTRA --steps $STEPS --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2>/dev/null )
find /tmp/kt_$CFG$SFX -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD" \
  "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS TCC_HIT_sum TCC_MISS_sum" \
  "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_VALU" \
  "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY" \
  "SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_ACTIVE_INST_EXP_GDS" \
  "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --kernel-include-regex "$KRE" --output-format csv --pmc $set -d /tmp/pmc_${CFG}${SFX}_$i -o pmc -- python $OLDPWD/bench.py --config $CFG $EXTRA --steps $PSTEPS --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$i.err )
  find /tmp/pmc_${CFG}${SFX}_$i -name "*counter_collection.csv" -exec cp {} $OUT/pmc_$i.csv \;
  gzip -f $OUT/pmc_$i.csv            # gpurun copies at most 64 MiB back
done
ls -la $OUT
