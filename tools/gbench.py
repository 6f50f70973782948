"""Timing of the photon gather alone (development aid): water.obj + Cornell box, photons emitted on
the host, queries = random points on the box walls.  RT_GATHER_PROF=1 prints the per-phase cycle split."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
torch.cuda.init()
import raytracer_amd as rt  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "data")


def main():
    nemit = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    s = rt.Scene()
    s.load_obj(os.path.join(DATA, "water.obj"), rgbs=(0.26, 0.84, 1.0, 1.8), ior=1.33)
    s.add_cornell()
    s.build_bvh()
    ctx = rt.Context(0)
    ctx.upload(s)
    s.emit_photons(nemit, seed=12345)
    ph = s.photons_get(0)
    ctx.photons_upload(ph)
    rng = np.random.default_rng(1)
    coherent = os.environ.get("GBENCH_COHERENT", "1") == "1"
    if coherent:
        # raster order over the five walls, like the requests of a frame (neighbouring queries are close)
        side = int(np.sqrt(nq / 5))
        u, v = np.meshgrid(np.linspace(-0.98, 0.98, side), np.linspace(-0.98, 0.98, side), indexing="ij")
        nq = 5 * side * side
        pos = np.zeros((nq, 3)); face = np.repeat(np.arange(5), side * side)
        uu, vv = np.tile(u.ravel(), 5), np.tile(v.ravel(), 5)
    else:
        pos = rng.uniform(-1, 1, (nq, 3))
        face = rng.integers(0, 5, nq)
    nrm = np.zeros((nq, 3))
    for f, (ax, val) in enumerate([(1, -1.0), (1, 1.0), (2, -1.0), (2, 1.0), (0, -1.0)]):
        m = face == f
        if coherent:
            others = [a for a in range(3) if a != ax]
            pos[m, others[0]] = uu[m]; pos[m, others[1]] = vv[m]
        pos[m, ax] = val * 0.999
        nrm[m, ax] = -val
    col = np.full((nq, 3), 0.8)
    for k in (1000, 100):
        p = rt.default_params(photon_num=nemit, photon_use=k)
        ctx.knn_estimate(p, pos[:1000], nrm[:1000], col[:1000])
        t0 = time.perf_counter()
        ctx.knn_estimate(p, pos, nrm, col)
        dt = time.perf_counter() - t0
        print(f"k={k}: {nq} queries over {len(ph)} photons in {dt * 1e3:.1f} ms (host wall incl. copies) -> {nq / dt / 1e6:.2f} Mq/s")


if __name__ == "__main__":
    main()
