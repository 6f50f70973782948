#!/bin/bash
# the headline (bench.py, 300 steps) with the library in the tree and with each ab/librt_amd_<name>.so given, twice round (scratch copy of the tree)
# (ab/librt_amd_<name>.so: other builds of the library, e.g. `hipcc ... -DRT_SOMETHING -c rt_device.hip` linked like the Makefile does; ab/ is ignored by git)
cd ${GRAFT_REPO_ROOT:-.}
cp raytracer_amd/librt_amd.so /tmp/librt_amd_tree.so
for rep in 1 2; do
  for name in tree "$@"; do
    cp $([ $name = tree ] && echo /tmp/librt_amd_tree.so || echo ab/librt_amd_$name.so) raytracer_amd/librt_amd.so
    echo -n "$name: "
    timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-extras --no-cpu-baseline --no-two-ctx 2> /tmp/err.txt | python3 -c "import json,sys; b=json.loads(sys.stdin.read()); print(b['value'], b['ms_per_step'], 'f32', b.get('alt_precision',{}).get('value'), 'parity ok', b['parity']['ok'], b['parity']['shadow_ok'])" || { tail -3 /tmp/err.txt; exit 1; }
  done
done
