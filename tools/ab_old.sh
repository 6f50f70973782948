#!/bin/bash
# A/B on the GPU box: the water frame (k = C4_K) with the library in the tree and with ab/librt_amd_old.so in its place, alternating (scratch copy of the tree)
# (ab/librt_amd_old.so: the library built from another commit, e.g. in a `git worktree`; ab/ is ignored by git)
cd ${GRAFT_REPO_ROOT:-.}
cp raytracer_amd/librt_amd.so /tmp/librt_amd_new.so
for rep in 1 2 3; do
  for which in new old; do
    cp $([ $which = old ] && echo ab/librt_amd_old.so || echo /tmp/librt_amd_new.so) raytracer_amd/librt_amd.so
    echo -n "$which: "
    PMC_REPS=4 timeout -k 10 120 python3 tools/pmc_run.py c4 2>&1 | grep "^c4" | sed 's/.*device//' || exit 1
  done
done
