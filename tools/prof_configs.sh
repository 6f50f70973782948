#!/bin/bash
# kernel-level time split of the C3/C4 renders (run through gpurun from the repo root)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_configs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/measure_configs.py > $OUT/configs.log 2> $OUT/trace.err || exit 1
cat $OUT/trace/*/*kernel_stats.csv | cut -c1-60,200-400
