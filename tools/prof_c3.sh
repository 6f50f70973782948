#!/bin/bash
# kernel-level time split of the 4K glass frame, with the queue tracer's scene in LDS and without (run through gpurun from the repo root)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_c3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lds -- python3 $ROOT/tools/measure_configs.py c3 > $OUT/lds.log 2> $OUT/lds.err || exit 1
export RT_NO_SCENE_LDS=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/l1 -- python3 $ROOT/tools/measure_configs.py c3 > $OUT/l1.log 2> $OUT/l1.err || exit 1
for v in lds l1; do echo "== $v"; cat $OUT/$v.log; cat $OUT/$v/*/*kernel_stats.csv | cut -c1-70,200-330; done
