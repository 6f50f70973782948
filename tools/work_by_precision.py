"""Node visits and primitive tests of the headline frame by precision and shadow mode (development aid): do the float walks
do more work than the double ones, or the same work more slowly?"""
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
buf = torch.empty((2, h, w), dtype=torch.int32, device="cuda")
for prec, name in ((rt.RT_F64, "f64"), (rt.RT_F32, "f32")):
    for shadow in (0, 1, 2):
        p = rt.default_params(width=w, height=h, precision=prec, shadow_rays=shadow)
        ms = []
        for i in range(30):
            st = ctx.trace_primary_dev(cam, p, (0, h), buf[0].data_ptr(), None, buf[1].data_ptr() if shadow == 1 else None)
            if i >= 10:
                ms.append(st.trace_kernel_ms)
        pc = rt.default_params(width=w, height=h, precision=prec, shadow_rays=shadow, count_work=1)
        sc = ctx.trace_primary_dev(cam, pc, (0, h), buf[0].data_ptr(), None, buf[1].data_ptr() if shadow == 1 else None)
        rays = st.primary_rays + st.shadow_rays
        print(f"{name} shadow={shadow}: {np.median(ms):.4f} ms, {rays} rays, inner visits {sc.inner_visits} ({sc.inner_visits / rays:.2f}/ray), primitive tests {sc.prim_tests} ({sc.prim_tests / rays:.2f}/ray)", flush=True)
