"""Secondary-ray cost of the glass teapot frame by recursion depth (development aid)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
torch.cuda.init()
import raytracer_amd as rt  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "data")
ctx = rt.Context(0)
s = rt.Scene()
s.load_obj(os.path.join(DATA, "teapot_normal.obj"), (1.0, 1.0, 1.0, 1.8), 1.5, (0, -0.7, 0))
s.add_cornell(); s.build_bvh()
cam = rt.look_at((2.8, 0, 0), (0, 0, 0))
for tree in ("reference", "device"):
    ctx.upload(s) if tree == "reference" else ctx.upload_lbvh(s)
    prev = None
    for depth in (1, 2, 3, 4):
        p = rt.default_params(width=3840, height=1088, use_photon_map=0, tracing_depth=depth)
        best = None
        for _ in range(3):
            rgb, bgr, st = ctx.render(cam, p)
            best = st if best is None or st.kernel_ms < best.kernel_ms else best
        extra = "" if prev is None else f"  (+{best.secondary_rays - prev[0]} rays, +{(best.kernel_ms - prev[1]) * 1e3:.0f} us -> {(best.secondary_rays - prev[0]) / max(best.kernel_ms - prev[1], 1e-9) / 1e3:.0f} Mrays/s)"
        print(f"{tree} tree, depth {depth}: {best.secondary_rays} secondary rays, device {best.kernel_ms:.3f} ms{extra}", flush=True)
        prev = (best.secondary_rays, best.kernel_ms)
