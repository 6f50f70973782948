"""Subsurface teapot frame (the reference's default material), device time (development aid)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import raytracer_amd as rt
DATA = os.path.join(ROOT, "tests", "golden", "data")
s = rt.Scene()
s.load_obj(os.path.join(DATA, "teapot_normal.obj"), (0.9, 0.7, 0.6, -1.0), 1.3, (0, -0.7, 0))
s.add_cornell(); s.build_bvh()
ctx = rt.Context(0); ctx.upload(s)
n_emit = int(os.environ.get("SSS_EMIT", "200000"))
(ns, nv), est = ctx.emit_photon_maps(n_emit, seed=12345)
print(f"photons: {ns} surface, {nv} volume")
cam = rt.look_at((2.8, 0, 0), (0, 0, 0))
for k in [int(x) for x in os.environ.get("SSS_KS", "1000,100").split(",")]:
    p = rt.default_params(width=1920, height=1080, photon_num=n_emit, photon_use=k)
    best = None
    for _ in range(3):
        rgb, bgr, st = ctx.render(cam, p)
        best = st if best is None or st.kernel_ms < best.kernel_ms else best
    print(f"subsurface teapot 1920x1080 k={k}: {best.gather_queries} gathers ({best.gather_single} one per wave), device {best.kernel_ms:.2f} ms (trace+shade {best.trace_kernel_ms:.2f}), mean {rgb.mean():.5f}", flush=True)
