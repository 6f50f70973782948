"""Device time of the photon emission (water + Cornell box, 1 M photons): kernels of the walk as the library's events see
them, and the wall time of the whole call with ordering and both trees (development aid; RT_EMIT_RESORT=1, RT_EMIT_UNORDERED=1)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import raytracer_amd as rt
DATA = os.path.join(ROOT, "tests", "golden", "data")
s = rt.Scene(); s.load_obj(os.path.join(DATA, "water.obj"), (0.26, 0.84, 1.0, 1.8), 1.33, (0, 0, 0)); s.add_cornell(); s.build_bvh()
ctx = rt.Context(0); ctx.upload(s)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
res = []
for i in range(6):
    t0 = time.perf_counter()
    counts, st = ctx.emit_photon_maps(n, seed=12345)
    ctx.synchronize()
    res.append((st.trace_kernel_ms, (time.perf_counter() - t0) * 1e3))
pos = ctx.photon_map_get(0)
print(f"{n} photons emitted, {counts[0]} + {counts[1]} stored, {st.secondary_rays} rays; emission kernels {min(r[0] for r in res[1:]):.3f} ms, whole call {min(r[1] for r in res[1:]):.2f} ms; "
      f"checksum of the surface map {np.sort(pos[:, 0]).sum():.9f} {pos[:, 3:6].sum():.9f}")
