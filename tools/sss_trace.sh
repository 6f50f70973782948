#!/bin/bash
# kernel table of the subsurface teapot frame under rocprofv3's kernel trace: bash tools/sss_trace.sh <tag>
TAG=${1:-sss}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/sss_bench.py > $OUT/trace.out 2> $OUT/trace.err || { echo "kernel trace failed"; tail -5 $OUT/trace.err; exit 1; }
python3 $ROOT/tools/c4_timeline.py $OUT/trace
