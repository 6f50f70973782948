"""Two f64 primary-ray frames of the instanced grid (10 000 teapots, 7680x4320), for counter collection under rocprofv3."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
torch.cuda.init()
import raytracer_amd as rt  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "data")
side = int(os.environ.get("GRID_SIDE", "100"))
ctx = rt.Context(0)
s = rt.Scene()
for i in range(side):
    for j in range(side):
        s.load_obj(os.path.join(DATA, "teapot_normal.obj"), (0.5, 0.5, 0.5, 0.0), 1.0, (0.0, i * 1.5, j * 1.5))
s.build_bvh()
ctx.upload(s)
c = side * 1.5 / 2
cam = rt.look_at((side * 1.5 * 0.6, c, c), (0.0, c, c))
w, h = 7680, 4320
buf = torch.empty((2, h, w), dtype=torch.int32, device="cuda")
p = rt.default_params(width=w, height=h, precision=rt.RT_F64, shadow_rays=0)
for _ in range(2):
    st = ctx.trace_primary_dev(cam, p, (0, h), buf[0].data_ptr(), None, None)
print(f"{st.trace_kernel_ms:.2f} ms", flush=True)
