"""Kernel-level timing of the closest-hit path on one GPU (development aid):
trace_primary on the bench frame, both instantiations, with / without shadow rays."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (first: torch carries its own HIP runtime and must be the one that initialises)
torch.cuda.init()
import raytracer_amd as rt  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "data")


def main():
    mesh = sys.argv[1] if len(sys.argv) > 1 else "teapot_normal.obj"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    s = rt.Scene(); s.load_obj(os.path.join(DATA, mesh)); s.add_cornell(); s.build_bvh()
    ctx = rt.Context(0); ctx.upload(s)
    cam = rt.look_at((2.8, 0, 0), (0, 0, 0))
    w, h = 1920, 1080
    buf = torch.empty((2, h, w), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for prec, name in ((rt.RT_F64, "f64"), (rt.RT_F32, "f32")):
        for shadow in (0, 1):
            p = rt.default_params(width=w, height=h, precision=prec, shadow_rays=shadow)
            ms = []
            for i in range(reps + 3):
                st = ctx.trace_primary_dev(cam, p, (0, h), buf[0].data_ptr(), None, buf[1].data_ptr())
                if i >= 3:
                    ms.append(st.trace_kernel_ms)
            rays = st.primary_rays + st.shadow_rays
            print(f"{mesh} {name} shadow={shadow}: min {min(ms):.4f} ms  median {np.median(ms):.4f} ms  -> {rays / np.median(ms) / 1e3:.0f} Mrays/s")


if __name__ == "__main__":
    main()
