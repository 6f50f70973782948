"""The 10 M-triangle frame at 16 samples per pixel, device time only (development aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import raytracer_amd as rt
DATA = os.path.join(ROOT, "tests", "golden", "data")
n = int(os.environ.get("C5_SIDE", "100"))
s = rt.Scene()
for i in range(n):
    for j in range(n):
        s.load_obj(os.path.join(DATA, "teapot_normal.obj"), (0.5, 0.5, 0.5, 0.0), 1.0, (0.0, i * 1.5, j * 1.5))
s.build_bvh()
ctx = rt.Context(0)
ctx.upload(s)
c = n * 1.5 / 2
cam = rt.look_at((n * 1.5 * 0.6, c, c), (0.0, c, c))
w, h = 7680, 4320
bgr = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
for prec, name in ((rt.RT_F64, "f64"), (rt.RT_F32, "f32")):
    for aa in (4, 2, 1):
        p = rt.default_params(width=w, height=h, precision=prec, aadepth=aa, use_photon_map=0)
        best = None
        for _ in range(3):
            st = ctx.render_dev(cam, p, (0, h), None, bgr.data_ptr(), stats=True)
            best = st if best is None or st.kernel_ms < best.kernel_ms else best
        print(f"C5 {name} aadepth {aa}: {best.primary_rays} rays, {best.kernel_ms:.2f} ms, checksum {int(bgr.sum().item())}", flush=True)
