"""The camera-ray kernel under RT_TILE_STRATA = 1 .. 5 over 36 views (development aid): ms per frame, 150 frames each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import raytracer_amd as rt
DATA = os.path.join(ROOT, "tests", "golden", "data")
w, h, N = 1920, 1080, 150
cams = {"front": ((2.8, 0, 0), (0, 0, 0)), "above": ((2.0, 1.6, 0.8), (0, -0.3, 0)), "low": ((2.4, -0.8, -1.0), (0, -0.5, 0))}
wins = {}
for mesh in ("teapot1.obj", "teapot_normal.obj"):
    s = rt.Scene(); s.load_obj(os.path.join(DATA, mesh)); s.add_cornell(); s.build_bvh()
    ctx = rt.Context(0); ctx.upload(s)
    ids = torch.empty((h, w), dtype=torch.int32, device="cuda"); sh = torch.empty((h, w), dtype=torch.int32, device="cuda")
    for cname, (eye, at) in cams.items():
        cam = rt.look_at(eye, at)
        for prec, pname in ((rt.RT_F64, "f64"), (rt.RT_F32, "f32")):
            for shadow in (2, 0, 1):
                p = rt.default_params(width=w, height=h, precision=prec, shadow_rays=shadow)
                out = {}
                for k in (1, 2, 3, 4, 5):
                    os.environ["RT_TILE_STRATA"] = str(k)
                    for _ in range(20): ctx.trace_primary_dev(cam, p, (0, h), ids.data_ptr(), None, sh.data_ptr() if shadow == 1 else None, stats=False)
                    ctx.synchronize(); t0 = time.perf_counter()
                    for _ in range(N): ctx.trace_primary_dev(cam, p, (0, h), ids.data_ptr(), None, sh.data_ptr() if shadow == 1 else None, stats=False)
                    ctx.synchronize(); out[k] = (time.perf_counter() - t0) / N * 1e3
                best12 = min(out[1], out[2]); best = min(out, key=out.get)
                wins[best] = wins.get(best, 0) + 1
                print(f"{mesh:18s} {cname:6s} {pname} shadow {shadow}: " + "  ".join(f"{k}: {v:.4f}" for k, v in out.items()) + f"   best {best} ({(out[best] / best12 - 1) * 100:+.1f} % against the better of 1, 2)", flush=True)
    del ctx
print("views won by each number of strata:", wins)
