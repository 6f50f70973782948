"""The camera-ray kernel under two values of a tuning switch -- SWEEP_VAR (default RT_TILE_STRATA) = 1, 2 -- (development aid): ms per frame for the three kinds of shadow query, two
meshes, three cameras and both precisions, 200 frames each."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import raytracer_amd as rt
DATA = os.path.join(ROOT, "tests", "golden", "data")
w, h = 1920, 1080
N = 200
VAR = os.environ.get("SWEEP_VAR", "RT_TILE_STRATA")
cams = {"front": ((2.8, 0, 0), (0, 0, 0)), "above": ((2.0, 1.6, 0.8), (0, -0.3, 0)), "low": ((2.4, -0.8, -1.0), (0, -0.5, 0))}
for mesh in ("teapot1.obj", "teapot_normal.obj"):
    s = rt.Scene(); s.load_obj(os.path.join(DATA, mesh)); s.add_cornell(); s.build_bvh()
    ctx = rt.Context(0); ctx.upload(s)
    ids = torch.empty((h, w), dtype=torch.int32, device="cuda")
    sh = torch.empty((h, w), dtype=torch.int32, device="cuda")
    for cname, (eye, at) in cams.items():
        cam = rt.look_at(eye, at)
        for prec, pname in ((rt.RT_F64, "f64"), (rt.RT_F32, "f32")):
            for shadow in (2, 0, 1):
                p = rt.default_params(width=w, height=h, precision=prec, shadow_rays=shadow)
                out = []
                for strata in ("1", "2", None):      # (None: the library's own choice, measured on the view's first launches)
                    if strata is None:
                        os.environ.pop(VAR, None)
                    else:
                        os.environ[VAR] = strata
                    for _ in range(30): ctx.trace_primary_dev(cam, p, (0, h), ids.data_ptr(), None, sh.data_ptr() if shadow == 1 else None, stats=False)
                    ctx.synchronize(); t0 = time.perf_counter()
                    for _ in range(N): ctx.trace_primary_dev(cam, p, (0, h), ids.data_ptr(), None, sh.data_ptr() if shadow == 1 else None, stats=False)
                    ctx.synchronize(); out.append((time.perf_counter() - t0) / N * 1e3)
                print(f"{mesh:18s} {cname:6s} {pname} shadow {shadow}: {VAR}=1 {out[0]:.4f} ms, =2 {out[1]:.4f} ms ({(out[1] / out[0] - 1) * 100:+.1f} %), chosen by the library {out[2]:.4f} ms "
                      f"({(out[2] / min(out[0], out[1]) - 1) * 100:+.1f} % of the better)", flush=True)
    del ctx
