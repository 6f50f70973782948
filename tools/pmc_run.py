"""One configuration of bench.py, a few frames, nothing else: the program rocprofv3 runs (kernel trace, or one group of
counters per run) to time and count what the kernels of that configuration do (profiles/profile_config.sh <tag> <cfg>).
  c2  the headline: teapot1.obj + Cornell box, 1920x1080, camera rays + one occlusion query per hit (60 frames per
      precision, double then float), exactly the launches bench.py times
  c3  glass teapot 3840x2160, depth 4, no photon map        (render_primary / shade / trace_queue kernels)
  c4  water + Cornell box, 1 M photons emitted on the device, k = 1000, 1920x1080   (gather kernels)
  sss the subsurface teapot (the reference's default material) + Cornell box, 200 000 photons emitted on the device (surface
      and volume maps), k = 1000, 1920x1080   (gather kernels, subsurface instantiation)
  c5  the 10 M-triangle frame of bench.py: C5_INST teapots on a grid (default 10 000), the reference's tree built on the
      host (C5_TREE=lbvh: the device-built tree), 7680x4320, 16 samples per pixel, double precision (C5_PREC=f32: float).
      The host build takes 16 s, so the first run leaves the flat image of the scene in C5_IMAGE (default
      /tmp/c5_scene.img) and later runs upload that."""
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
    if which == "c2":
        s.load_obj(os.path.join(DATA, "teapot1.obj"))
        s.add_cornell(); s.build_bvh(); ctx.upload(s)
        w, h = 1920, 1080
        ids = torch.empty((h, w), dtype=torch.int32, device="cuda")
        for prec in (rt.RT_F64, rt.RT_F32):
            p = rt.default_params(width=w, height=h, precision=prec, shadow_rays=2)
            # (launches two to thirteen of a view are the library's timed trials of its two tile orders: a synchronisation behind
            # them, so that the rest run in the order it settled on, and enough launches that the trials weigh little)
            for i in range(max(reps, 60)):
                ctx.trace_primary_dev(CAM, p, (0, h), ids.data_ptr(), None, None, stats=False)
                if i == 14:
                    ctx.synchronize()
            ctx.synchronize()
        print("c2: %d launches per precision" % max(reps, 60))
        return
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
    elif which == "sss":
        s.load_obj(os.path.join(DATA, "teapot_normal.obj"), (0.9, 0.7, 0.6, -1.0), 1.3, (0, -0.7, 0))
        s.add_cornell(); s.build_bvh(); ctx.upload(s)
        ctx.emit_photon_maps(200000, seed=12345)
        w, h = 1920, 1080
        p = rt.default_params(width=w, height=h, photon_num=200000, photon_use=int(os.environ.get("C4_K", "1000")))
        cam = CAM
    else:
        n_inst = int(os.environ.get("C5_INST", "10000"))
        side = int(np.ceil(np.sqrt(n_inst)))
        lbvh = os.environ.get("C5_TREE", "ref") == "lbvh"
        image = os.environ.get("C5_IMAGE", "/tmp/c5_scene.img") + (".%d" % side)
        if not lbvh and os.path.exists(image):
            ctx.upload_flat(np.fromfile(image, np.uint8))
        else:
            for i in range(side):
                for j in range(side):
                    s.load_obj(os.path.join(DATA, "teapot_normal.obj"), (0.5, 0.5, 0.5, 0.0), 1.0, (0.0, i * 1.5, j * 1.5))
            if lbvh:
                ctx.upload_lbvh(s)
            else:
                s.build_bvh()
                img = s.export_flat()
                img.tofile(image + ".tmp")
                os.replace(image + ".tmp", image)
                ctx.upload_flat(img)
                del img
        c = side * 1.5 / 2
        cam = rt.look_at((side * 1.5 * 0.6, c, c), (0.0, c, c))
        w, h = 7680, 4320
        p = rt.default_params(width=w, height=h, aadepth=4, use_photon_map=0, precision=rt.RT_F32 if os.environ.get("C5_PREC") == "f32" else rt.RT_F64)
    bgr = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
    for _ in range(reps):
        st = ctx.render_dev(cam, p, (0, h), None, bgr.data_ptr())
    print(f"{which}: {st.primary_rays} primary + {st.secondary_rays} secondary rays, {st.gather_queries} gathers, device {st.kernel_ms:.3f} ms "
          f"(trace + shade {st.trace_kernel_ms:.3f} ms)")


if __name__ == "__main__":
    main()
