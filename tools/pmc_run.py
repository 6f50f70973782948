"""One configuration, a few frames, nothing else: the program rocprofv3 --pmc runs to count what the kernels of that
configuration do (profiles/run_pmc.sh <tag> python3 tools/pmc_run.py <c3|c4|c5>).
  c3  glass teapot 3840x2160, depth 4, no photon map        (render_primary / shade / trace_queue kernels)
  c4  water + Cornell box, 1 M photons emitted on the device, k = 1000, 1920x1080   (gather kernels)
  c5  N teapots on a grid (C5_INST, default 2500 = 2.5 M triangles), 7680x4320, 16 samples per pixel, device-built tree"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (first: torch carries its own HIP runtime and must be the one that initialises)
torch.cuda.init()
import raytracer_amd as rt  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "data")
CAM = rt.look_at((2.8, 0, 0), (0, 0, 0))


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c4"
    reps = int(os.environ.get("PMC_REPS", "3"))
    ctx = rt.Context(0)
    s = rt.Scene()
    if which == "c3":
        s.load_obj(os.path.join(DATA, "teapot_normal.obj"), (1.0, 1.0, 1.0, 1.8), 1.5, (0, -0.7, 0))
        s.add_cornell(); s.build_bvh(); ctx.upload(s)
        w, h = 3840, 2160
        p = rt.default_params(width=w, height=h, use_photon_map=0)
        cam = CAM
    elif which == "c4":
        s.load_obj(os.path.join(DATA, "water.obj"), (0.26, 0.84, 1.0, 1.8), 1.33, (0, 0, 0))
        s.add_cornell(); s.build_bvh(); ctx.upload(s)
        ctx.emit_photon_maps(1000000, seed=12345)
        w, h = 1920, 1080
        p = rt.default_params(width=w, height=h, photon_num=1000000, photon_use=int(os.environ.get("C4_K", "1000")))
        cam = CAM
    else:
        n_inst = int(os.environ.get("C5_INST", "2500"))
        side = int(np.ceil(np.sqrt(n_inst)))
        for i in range(side):
            for j in range(side):
                s.load_obj(os.path.join(DATA, "teapot_normal.obj"), (0.5, 0.5, 0.5, 0.0), 1.0, (0.0, i * 1.5, j * 1.5))
        ctx.upload_lbvh(s)
        c = side * 1.5 / 2
        cam = rt.look_at((side * 1.5 * 0.6, c, c), (0.0, c, c))
        w, h = 7680, 4320
        p = rt.default_params(width=w, height=h, aadepth=4, use_photon_map=0)
    bgr = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
    for _ in range(reps):
        st = ctx.render_dev(cam, p, (0, h), None, bgr.data_ptr())
    print(f"{which}: {st.primary_rays} primary + {st.secondary_rays} secondary rays, {st.gather_queries} gathers, device {st.kernel_ms:.3f} ms "
          f"(trace + shade {st.trace_kernel_ms:.3f} ms)")


if __name__ == "__main__":
    main()
