"""Throughput of the batch form of BVH::Shoot (rt_trace_closest_dev) on one GPU (development aid): camera rays in
raster order and rays with random origins and directions, on the teapot scene and on a grid of teapots."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
torch.cuda.init()
import raytracer_amd as rt  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "data")


def rays_camera(w, h, pos, target):
    cam = rt.look_at(pos, target)
    f = np.array([list(cam.dir), list(cam.right), list(cam.down)], dtype=np.float64) if hasattr(cam, "dir") else None
    x = (np.arange(w) + 0.5) / w
    y = ((h - np.arange(h)) + 0.5) / h
    asp = w / h
    X, Y = np.meshgrid((x - 0.5) * asp, y - 0.5)
    cd = np.array(target, float) - np.array(pos, float); cd /= np.linalg.norm(cd)
    cr = np.cross([0, 1, 0], cd); cr /= np.linalg.norm(cr)
    cdn = np.cross(cr, cd)
    d = cd[None, None, :] + cr[None, None, :] * X[..., None] + cdn[None, None, :] * Y[..., None]
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d = d.reshape(-1, 3)
    o = np.broadcast_to(np.array(pos, float), d.shape).copy()
    return o, d


def rays_random(n, lo, hi, seed=1):
    r = np.random.default_rng(seed)
    o = r.uniform(lo, hi, (n, 3))
    d = r.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


def run(ctx, name, o, d, reps=5):
    n = len(o)
    org = torch.tensor(np.ascontiguousarray(o.T), device="cuda")      # SoA x[n] y[n] z[n]
    dr = torch.tensor(np.ascontiguousarray(d.T), device="cuda")
    ids = torch.empty(n, dtype=torch.int32, device="cuda")
    ext = torch.cuda.ExternalStream(ctx.stream)
    for prec, pn in ((rt.RT_F64, "f64"), (rt.RT_F32, "f32")):
        best = 1e9
        for _ in range(reps):
            dist = torch.full((n,), 1e9, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ext)
            ctx.trace_closest_dev(n, org.data_ptr(), dr.data_ptr(), dist.data_ptr(), ids.data_ptr(), precision=prec)
            e1.record(ext)
            ctx.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print(f"{name} {pn}: {n} rays, {best:.3f} ms -> {n / best / 1e3:.0f} Mrays/s, {int((ids >= 0).sum())} hits, id sum {int(ids.to(torch.int64).sum())}", flush=True)


def main():
    ctx = rt.Context(0)
    s = rt.Scene(); s.load_obj(os.path.join(DATA, "teapot1.obj")); s.add_cornell(); s.build_bvh()
    ctx.upload(s)
    o, d = rays_camera(1920, 1080, (2.8, 0, 0), (0, 0, 0))
    run(ctx, "teapot1 camera rays", o, d)
    o, d = rays_random(2_000_000, -0.95, 0.95)
    run(ctx, "teapot1 random rays", o, d)
    side = int(os.environ.get("GRID_SIDE", "30"))
    s = rt.Scene()
    for i in range(side):
        for j in range(side):
            s.load_obj(os.path.join(DATA, "teapot_normal.obj"), (0.5, 0.5, 0.5, 0.0), 1.0, (0.0, i * 1.5, j * 1.5))
    s.build_bvh()
    ctx.upload(s)
    print(f"grid of {side * side} teapots: {s.size()} triangles, depth {s.bvh_depth()}")
    c = side * 1.5 / 2
    o, d = rays_camera(3840, 2160, (side * 1.5 * 0.6, c, c), (0.0, c, c))
    run(ctx, "grid camera rays", o, d)
    o, d = rays_random(2_000_000, -1.0, side * 1.5)
    o[:, 0] = np.random.default_rng(2).uniform(-3, 3, len(o))
    run(ctx, "grid random rays", o, d)


if __name__ == "__main__":
    main()
