#!/bin/bash
# A/B on the GPU box: the subsurface teapot frame with the library in the tree and with ab/librt_amd_old.so in its place, alternating
# (ab/librt_amd_old.so: the library built from another commit, e.g. in a `git worktree`; ab/ is ignored by git)
cd ${GRAFT_REPO_ROOT:-.}
cp raytracer_amd/librt_amd.so /tmp/librt_amd_new.so
for rep in 1 2 3; do
  for which in new old; do
    cp $([ $which = old ] && echo ab/librt_amd_old.so || echo /tmp/librt_amd_new.so) raytracer_amd/librt_amd.so
    timeout -k 10 120 python3 tools/sss_bench.py 2>&1 | grep "^subsurface" | sed "s/.*k=/$which k=/; s/gathers.*device/device/" || exit 1
  done
done
