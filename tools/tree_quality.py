"""Work per ray on the reference's tree and on the device-built one (inner-node visits, primitive
tests, kernel time) for the teapot in its box and for a grid of teapots."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
torch.cuda.init()
import raytracer_amd as rt  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "data")


def measure(ctx, cam, w, h, label):
    buf = torch.empty((h, w), dtype=torch.int32, device="cuda")
    for prec, name in ((rt.RT_F64, "f64"), (rt.RT_F32, "f32")):
        p = rt.default_params(width=w, height=h, precision=prec, shadow_rays=0)
        for _ in range(2):
            st = ctx.trace_primary_dev(cam, p, (0, h), buf.data_ptr())
        pc = rt.default_params(width=w, height=h, precision=prec, shadow_rays=0, count_work=1)
        sc = ctx.trace_primary_dev(cam, pc, (0, h), buf.data_ptr())
        n = st.primary_rays
        print(f"  {label} {name}: {st.trace_kernel_ms:.3f} ms, {sc.inner_visits / n:.1f} inner visits/ray, {sc.prim_tests / n:.1f} primitive tests/ray", flush=True)


def main():
    ctx = rt.Context(0)
    s = rt.Scene()
    s.load_obj(os.path.join(DATA, "teapot_normal.obj"), (0.5, 0.5, 0.5, 0.0), 1.0, (0, -0.7, 0))
    s.add_cornell(); s.build_bvh()
    cam = rt.look_at((2.8, 0, 0), (0, 0, 0))
    print("teapot in the box, 1920x1080")
    ctx.upload(s); measure(ctx, cam, 1920, 1080, f"reference tree (depth {s.bvh_depth()})")
    ms = ctx.upload_lbvh(s); measure(ctx, cam, 1920, 1080, f"device tree ({ms:.2f} ms build)")
    side = int(os.environ.get("GRID_SIDE", "20"))
    s = rt.Scene()
    for i in range(side):
        for j in range(side):
            s.load_obj(os.path.join(DATA, "teapot_normal.obj"), (0.5, 0.5, 0.5, 0.0), 1.0, (0.0, i * 1.5, j * 1.5))
    s.build_bvh()
    c = side * 1.5 / 2
    cam = rt.look_at((side * 1.5 * 0.6, c, c), (0.0, c, c))
    print(f"{side}x{side} teapots ({s.size()} triangles), 3840x2160")
    ctx.upload(s); measure(ctx, cam, 3840, 2160, f"reference tree (depth {s.bvh_depth()})")
    ms = ctx.upload_lbvh(s); measure(ctx, cam, 3840, 2160, f"device tree ({ms:.2f} ms build)")


if __name__ == "__main__":
    main()
