#!/bin/bash
# the water frame (1 M photons, k = 1000 and 100) through the group gather (FORMS: stream tile), with and without the cycle accounting
cd ${GRAFT_REPO_ROOT:-.}
for form in ${FORMS:-stream tile}; do
  for k in 1000 100; do
    echo "== form $form k $k"
    RT_GATHER_FORM=$form C4_K=$k PMC_REPS=4 timeout -k 10 120 python3 tools/pmc_run.py c4 2>&1 | grep "^c4" || exit 1
    RT_GATHER_FORM=$form C4_K=$k PMC_REPS=2 RT_GATHER_PROF=1 timeout -k 10 120 python3 tools/pmc_run.py c4 2>&1 | grep -v "^c4\|amdgpu.ids\|^group gather" | tail -3
  done
done
