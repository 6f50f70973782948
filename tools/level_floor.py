"""Fixed cost of a ray level (development aid): a frame in which only a handful of pixels see a mirror, so that the
queue tracer's launches have next to nothing to trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import raytracer_amd as rt
s = rt.Scene()
s.add_triangle((0.0, -0.02, -0.02), (0.0, 0.02, -0.02), (0.0, 0.0, 0.02), (1.0, 1.0, 1.0, 0.9), 1.5)     # a small mirror facing the camera
s.add_cornell(); s.build_bvh()
ctx = rt.Context(0); ctx.upload(s)
cam = rt.look_at((2.8, 0, 0), (0, 0, 0))
for depth in (1, 2, 3, 4):
    p = rt.default_params(width=3840, height=2160, use_photon_map=0, tracing_depth=depth)
    best = None
    for _ in range(4):
        rgb, bgr, st = ctx.render(cam, p)
        best = st if best is None or st.kernel_ms < best.kernel_ms else best
    print(f"depth {depth}: {best.secondary_rays} secondary rays, device {best.kernel_ms * 1e3:.0f} us", flush=True)
