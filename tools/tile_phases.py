"""Cycles a wave of the headline launch spends on its camera rays and on its shadow rays, by precision (needs a library
built with -DRT_TILE_END_TIMES; part = -7 asks the kernel for the phase cycles instead of the end times)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import raytracer_amd as rt
DATA = os.path.join(ROOT, "tests", "golden", "data")
s = rt.Scene(); s.load_obj(os.path.join(DATA, "teapot1.obj")); s.add_cornell(); s.build_bvh()
ctx = rt.Context(0); ctx.upload(s)
cam = rt.look_at((2.8, 0, 0), (0, 0, 0))
w, h = 1920, 1080
ids = torch.empty((h, w), dtype=torch.int32, device="cuda")
t = torch.empty((h, w), dtype=torch.float64, device="cuda")
for prec, name in ((rt.RT_F64, "f64"), (rt.RT_F32, "f32")):
    p = rt.default_params(width=w, height=h, shadow_rays=2, precision=prec, part=-7)
    for i in range(5):
        st = ctx.trace_primary_dev(cam, p, (0, h), ids.data_ptr(), t.data_ptr(), None)
    ctx.synchronize()
    T = t.cpu().numpy()
    camc, shc = T[::8, 0::8], T[::8, 1::8]       # one value per tile (wave)
    q = lambda a: ", ".join(f"{k}%: {np.percentile(a, k):.0f}" for k in (10, 50, 90, 99, 100))
    print(f"{name}: kernel {st.trace_kernel_ms:.4f} ms; camera phase cycles per wave: mean {camc.mean():.0f} ({q(camc)}); shadow phase: mean {shc.mean():.0f} ({q(shc)})")
