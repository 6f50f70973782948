#!/bin/bash
# kernel table and last frame's timeline of the water frame (1 M photons, k = C4_K) under rocprofv3's kernel trace: bash tools/c4_trace.sh <tag>
TAG=${1:-c4trace}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/pmc_run.py c4 > $OUT/trace.out 2> $OUT/trace.err || { echo "kernel trace failed"; tail -5 $OUT/trace.err; exit 1; }
python3 $ROOT/tools/c4_timeline.py $OUT/trace
