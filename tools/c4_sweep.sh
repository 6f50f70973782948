#!/bin/bash
# the water frame's gather under one tuning switch at a time: bash tools/c4_sweep.sh VAR v1 v2 ...   (C4_K from the environment)
cd ${GRAFT_REPO_ROOT:-.}
VAR=$1; shift
for v in "$@"; do
  echo "== $VAR=$v k=${C4_K:-1000}"
  env $VAR=$v PMC_REPS=4 timeout -k 10 120 python3 tools/pmc_run.py c4 2>&1 | grep "^c4" || exit 1
done
