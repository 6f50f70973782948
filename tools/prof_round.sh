#!/bin/bash
# kernel tables and counters of the four bench configurations, then the bench itself: bash tools/prof_round.sh <tag>   (through gpurun, from the repo root)
TAG=${1:-r03c}
cd ${GRAFT_REPO_ROOT:-.}
for cfg in c2 c3 c4 c5; do
  bash profiles/profile_config.sh $TAG $cfg > gpurun_out/prof_${TAG}_${cfg}.log 2>&1 || { echo "profile of $cfg failed"; tail -3 gpurun_out/prof_${TAG}_${cfg}.log; exit 1; }
  echo "profiled $cfg"
done
timeout -k 10 600 python3 bench.py > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err || { echo "bench failed"; tail -5 gpurun_out/bench_${TAG}.err; exit 1; }
echo "bench done"
