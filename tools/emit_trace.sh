#!/bin/bash
# kernel table of the photon emission under rocprofv3's kernel trace: bash tools/emit_trace.sh <tag>   (RT_EMIT_RESORT from the environment)
TAG=${1:-emit}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/emit_bench.py > $OUT/trace.out 2> $OUT/trace.err || { echo "kernel trace failed"; tail -5 $OUT/trace.err; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "emit_kernel" in r["Name"] or "DeviceRadixSort" in r["Name"] or "radix" in r["Name"]:
        print(f"{r['Name'].split('(')[0][-60:]:62s} calls {r['Calls']:>5s} total us {float(r['TotalDurationNs'])/1e3:10.1f} avg {float(r['AverageNs'])/1e3:8.1f}")
PY
