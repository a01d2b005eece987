#!/bin/bash
# GPU box: ad-hoc PMC passes (counters only: no tracing flags beside --pmc) for one bench configuration and one kernel-name regex.
#   tools/pmc_probe.sh <cfg> <kernel regex> "<bench flags>" "<counter set 1>" ["<counter set 2>" ...]   -> prints per-kernel means
set -u
CFG=$1; KRE=$2; EXTRA=$3; shift 3
export TMPDIR=/tmp DMX_EXPERIMENTS=1 DMX_NO_OVERLAP=1
ROOT=$PWD; i=0
for set in "$@"; do
  i=$((i+1)); rm -rf /tmp/pp_$i
  ( cd /tmp && rocprofv3 --kernel-include-regex "$KRE" --output-format csv --pmc $set -d /tmp/pp_$i -o pmc -- python $ROOT/bench.py --config $CFG $EXTRA --only --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/pp_$i.err )
  python3 - /tmp/pp_$i <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n =", len(next(iter(cs.values()))))
PY
done
