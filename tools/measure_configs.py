"""Timing of the wider configurations on one GPU (informational; bench.py is the contract):
C3-like 4K glass teapot with the bounce queue, C4 water.obj + 1M emitted photons with the gather."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "data")
CAM = rt.look_at((2.8, 0, 0), (0, 0, 0))


def timed_render(ctx, p, reps=3):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        rgb, bgr, st = ctx.render(CAM, p)
        dt = time.perf_counter() - t0
        if best is None or st.kernel_ms < best[0].kernel_ms:
            best = (st, dt)
    return best


def main():
    which = sys.argv[1:] or ["c3", "c4"]
    ctx = rt.Context(0)
    if "c3" in which:
        s = rt.Scene()
        s.load_obj(os.path.join(DATA, "teapot_normal.obj"), (1.0, 1.0, 1.0, 1.8), 1.5, (0, -0.7, 0))
        s.add_cornell(); s.build_bvh()
        ctx.upload(s)
        for prec, name in ((rt.RT_F64, "f64"), (rt.RT_F32, "f32")):
            p = rt.default_params(width=3840, height=2160, use_photon_map=0, precision=prec)
            st, dt = timed_render(ctx, p)
            rays = st.primary_rays + st.secondary_rays
            print(f"C3 {name} 3840x2160 glass teapot, no photon map: {rays} rays, device {st.kernel_ms:.2f} ms -> {rays / st.kernel_ms / 1e3:.1f} Mrays/s (host wall {dt*1e3:.1f} ms)")
    if "c4" in which:
        nemit = int(os.environ.get("C4_EMIT", "1000000"))
        s = rt.Scene()
        s.load_obj(os.path.join(DATA, "water.obj"), (0.26, 0.84, 1.0, 1.8), 1.33, (0, 0, 0))
        s.add_cornell(); s.build_bvh()
        t0 = time.perf_counter()
        n = s.emit_photons(nemit, seed=12345)
        t1 = time.perf_counter()
        ph = s.photons_get()
        ctx.upload(s)
        ctx.photons_upload(ph)
        t2 = time.perf_counter()
        print(f"C4 host: emitted {nemit} -> {n} photons in {t1 - t0:.2f} s; kd build + upload {t2 - t1:.2f} s")
        for k in (1000, 100):
            p = rt.default_params(width=1920, height=1080, photon_num=nemit, photon_use=k)
            st, dt = timed_render(ctx, p, reps=2)
            rays = st.primary_rays + st.secondary_rays
            print(f"C4 water 1920x1080 k={k}: {rays} rays, {st.gather_queries} gathers, device {st.kernel_ms:.2f} ms "
                  f"(trace+shade {st.trace_kernel_ms:.2f} ms) -> {st.gather_queries / max(st.kernel_ms - st.trace_kernel_ms, 1e-9) / 1e3:.2f} Mqueries/s, host wall {dt*1e3:.0f} ms")


if __name__ == "__main__":
    main()
