"""Timing of the wider configurations on one GPU (informational; bench.py is the contract):
C3-like 4K glass teapot with the bounce queue, C4 water.obj + 1M emitted photons with the gather."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (first: torch carries its own HIP runtime and must be the one that initialises)
torch.cuda.init()
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
        s.photons_clear()
        t0 = time.perf_counter()
        est = s.emit_photons_gpu(ctx, nemit, seed=12345)
        t1 = time.perf_counter()
        print(f"C4 GPU emission: {nemit} -> {s.photon_count(0)} photons, {est.secondary_rays} rays, kernels {est.kernel_ms:.2f} ms, "
              f"wall incl. download and ordering {t1 - t0:.3f} s")
        ph = s.photons_get()
        for name, fn in (("host", ctx.photons_upload), ("device", ctx.photons_upload_dev_build)):
            fn(ph)
            t0 = time.perf_counter()
            fn(ph)
            print(f"C4 kd-tree of {len(ph)} photons, {name} build incl. upload: {(time.perf_counter() - t0) * 1e3:.1f} ms")
        for _ in range(2):
            t0 = time.perf_counter()
            (ns, nv), est = ctx.emit_photon_maps(nemit, seed=12345)
            t1 = time.perf_counter()
        print(f"C4 emission + ordering + kd-trees on the device: {ns} photons, kernels {est.kernel_ms:.2f} ms (emission {est.trace_kernel_ms:.2f}), wall {(t1 - t0) * 1e3:.1f} ms")
        for mode in ("group", "query"):
            os.environ["RT_GATHER_MODE"] = mode
            for k in [int(x) for x in os.environ.get("C4_K", "1000,100").split(",")]:
                p = rt.default_params(width=1920, height=1080, photon_num=nemit, photon_use=k)
                st, dt = timed_render(ctx, p, reps=2)
                rays = st.primary_rays + st.secondary_rays
                print(f"C4 water 1920x1080 k={k} gather={mode}: {rays} rays, {st.gather_queries} gathers ({st.gather_single} one per wave), device {st.kernel_ms:.2f} ms "
                      f"(trace+shade {st.trace_kernel_ms:.2f} ms) -> {st.gather_queries / max(st.kernel_ms - st.trace_kernel_ms, 1e-9) / 1e3:.2f} Mqueries/s, host wall {dt*1e3:.0f} ms")
        os.environ.pop("RT_GATHER_MODE")
    if "c5" in which:
        n_inst = int(os.environ.get("C5_INST", "10000"))
        side = int(np.ceil(np.sqrt(n_inst)))
        s = rt.Scene()
        t0 = time.perf_counter()
        k = 0
        for i in range(side):
            for j in range(side):
                if k < n_inst:
                    s.load_obj(os.path.join(DATA, "teapot_normal.obj"), (0.5, 0.5, 0.5, 0.0), 1.0, (0.0, i * 1.5, j * 1.5))
                    k += 1
        t1 = time.perf_counter()
        s.build_bvh()
        t2 = time.perf_counter()
        ctx.upload(s)
        t3 = time.perf_counter()
        print(f"C5 host: {s.size()} triangles loaded in {t1 - t0:.1f} s, BVH (depth {s.bvh_depth()}) built in {t2 - t1:.1f} s, upload {t3 - t2:.1f} s", flush=True)
        c = side * 1.5 / 2
        cam = rt.look_at((side * 1.5 * 0.6, c, c), (0.0, c, c))
        w, h = 7680, 4320
        buf = torch.empty((2, h, w), dtype=torch.int32, device="cuda")
        for prec, name in ((rt.RT_F64, "f64"), (rt.RT_F32, "f32")):
            for shadow in (0, 1):
                p = rt.default_params(width=w, height=h, precision=prec, shadow_rays=shadow, light_pos=(side * 1.5, c, c))
                for _ in range(2):
                    st = ctx.trace_primary_dev(cam, p, (0, h), buf[0].data_ptr(), None, buf[1].data_ptr())
                rays = st.primary_rays + st.shadow_rays
                hits = int((buf[0] >= 0).sum().item())
                print(f"C5 {name} {w}x{h} shadow={shadow}: {rays} rays ({hits} primary hits), kernel {st.trace_kernel_ms:.2f} ms -> {rays / st.trace_kernel_ms / 1e3:.0f} Mrays/s", flush=True)
        # the full pixel path at 16 samples per pixel (aadepth 4: the reference's pattern has 4 distinct rays, each weighted 4)
        rgb8 = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
        for prec, name in ((rt.RT_F64, "f64"), (rt.RT_F32, "f32")):
            p = rt.default_params(width=w, height=h, precision=prec, aadepth=4, use_photon_map=0)
            for _ in range(2):
                st = ctx.render_dev(cam, p, (0, h), None, rgb8.data_ptr())
            print(f"C5 {name} {w}x{h} pixel path, aadepth 4: {st.primary_rays} distinct camera rays (x4 = the reference's 16 per pixel), "
                  f"device {st.kernel_ms:.1f} ms -> {st.primary_rays / st.kernel_ms / 1e3:.0f} Mrays/s traced, {4 * st.primary_rays / st.kernel_ms / 1e3:.0f} Msamples/s", flush=True)
        del rgb8
        ref_ids = buf[0].clone()
        t0 = time.perf_counter()
        ms = ctx.upload_lbvh(s)
        t1 = time.perf_counter()
        print(f"C5 device-built tree (linear BVH): {ms:.1f} ms of kernels, {t1 - t0:.2f} s wall with the upload", flush=True)
        for prec, name in ((rt.RT_F64, "f64"), (rt.RT_F32, "f32")):
            p = rt.default_params(width=w, height=h, precision=prec, shadow_rays=0)
            for _ in range(2):
                st = ctx.trace_primary_dev(cam, p, (0, h), buf[0].data_ptr(), None, buf[1].data_ptr())
            diff = float((buf[0] != ref_ids).float().mean().item()) if prec == rt.RT_F32 else float("nan")
            print(f"C5 {name} on the device-built tree: kernel {st.trace_kernel_ms:.2f} ms -> {st.primary_rays / st.trace_kernel_ms / 1e3:.0f} Mrays/s"
                  + (f" (ids differing from the reference tree's f32 frame: {diff:.2e})" if prec == rt.RT_F32 else ""), flush=True)


if __name__ == "__main__":
    main()
