#!/bin/bash
# one group of counters over one configuration, summed per kernel: bash tools/pmc_one.sh <tag> <cfg> COUNTER...
TAG=$1; CFG=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/p -- python3 $ROOT/tools/pmc_run.py $CFG > $OUT/p.out 2> $OUT/p.err || { echo "pass failed"; tail -5 $OUT/p.err; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    if "gather" in k or "trace" in k or "render" in k or "emit" in k: print(f"{k:42s} {c:28s} {len(v):4d} {sum(v)/len(v):14.6g}")
PY
