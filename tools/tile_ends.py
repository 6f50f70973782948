"""When the waves of the headline launch finish (needs a library built with -DRT_TILE_END_TIMES: the t image then holds
the 100 MHz clock at which each pixel's wave was done).  Prints the distribution and a coarse map of the end times."""
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
p = rt.default_params(width=w, height=h, shadow_rays=2)
for i in range(5):
    st = ctx.trace_primary_dev(cam, p, (0, h), ids.data_ptr(), t.data_ptr(), None)
ctx.synchronize()
T = t.cpu().numpy()[::8, ::8]           # one value per tile
T = (T - T.min()) / 100.0               # us
print(f"kernel {st.trace_kernel_ms:.4f} ms; tiles {T.size}; end times us: " + ", ".join(f"{q}%: {np.percentile(T, q):.0f}" for q in (10, 25, 50, 75, 90, 95, 99, 100)))
# coarse map: mean end time per 16 x 8 tile super-block
hb, wb = T.shape[0] // 8, T.shape[1] // 16
M = T[:hb * 8, :wb * 16].reshape(hb, 8, wb, 16).mean(axis=(1, 3))
for r in M: print(" ".join(f"{v:4.0f}" for v in r))
