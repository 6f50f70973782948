#!/bin/bash
# the round's closing measurements in one GPU call: kernel stats + PMC of the default bench, PMC of the glass frame,
# the bench line at its default length and at 20 steps
set -u
TAG=${1:-r02h}
cd $GRAFT_REPO_ROOT
timeout -k 10 400 bash profiles/run_profile.sh $TAG > gpurun_out/prof_$TAG.log 2>&1 || { echo profile failed; exit 1; }
timeout -k 10 300 bash profiles/run_pmc.sh c3_$TAG python3 $GRAFT_REPO_ROOT/tools/pmc_run.py c3 > gpurun_out/pmc_c3_$TAG.log 2>&1 || { echo pmc c3 failed; exit 1; }
timeout -k 10 300 bash profiles/run_pmc.sh c4_$TAG python3 $GRAFT_REPO_ROOT/tools/pmc_run.py c4 > gpurun_out/pmc_c4_$TAG.log 2>&1 || { echo pmc c4 failed; exit 1; }
timeout -k 10 300 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { echo bench failed; exit 1; }
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_20.json 2> gpurun_out/bench_${TAG}_20.err || { echo bench20 failed; exit 1; }
tail -c 300 gpurun_out/bench_$TAG.json
