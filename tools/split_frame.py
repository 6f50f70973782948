"""The headline frame split over two streams of one GPU (development aid): each frame's upper and lower bands of rows are
traced by two contexts at once, and both streams wait for both halves before the next frame begins -- against one launch per frame."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import raytracer_amd as rt
DATA = os.path.join(ROOT, "tests", "golden", "data")
dev = torch.device("cuda", 0)
s = rt.Scene(); s.load_obj(os.path.join(DATA, "teapot1.obj")); s.add_cornell(); s.build_bvh()
cam = rt.look_at((2.8, 0, 0), (0, 0, 0))
w, h = 1920, 1080
N = int(os.environ.get("STEPS", "300"))
ctxs = [rt.Context(0), rt.Context(0)]
for c in ctxs: c.upload(s)
exts = [torch.cuda.ExternalStream(c.stream, device=dev) for c in ctxs]
# one launch per frame
p1 = rt.default_params(width=w, height=h, shadow_rays=2)
full = torch.empty((h, w), dtype=torch.int32, device=dev)
for _ in range(50): ctxs[0].trace_primary_dev(cam, p1, (0, h), full.data_ptr(), None, None, stats=False)
ctxs[0].synchronize(); t0 = time.perf_counter()
for _ in range(N): ctxs[0].trace_primary_dev(cam, p1, (0, h), full.data_ptr(), None, None, stats=False)
ctxs[0].synchronize(); one = (time.perf_counter() - t0) / N * 1e3
for parts in (2, 4, 8):
    cut = [(h // 8 * i // parts) * 8 for i in range(parts + 1)]      # contiguous bands of rows, multiples of 8
    bufs = [torch.empty((cut[i + 1] - cut[i], w), dtype=torch.int32, device=dev) for i in range(parts)]
    def frame():
        for i in range(parts):
            ctxs[i & 1].trace_primary_dev(cam, p1, (cut[i], cut[i + 1]), bufs[i].data_ptr(), None, None, stats=False)
        ev = [torch.cuda.Event(), torch.cuda.Event()]
        ev[0].record(exts[0]); ev[1].record(exts[1])
        exts[0].wait_event(ev[1]); exts[1].wait_event(ev[0])
    for _ in range(50): frame()
    for c in ctxs: c.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): frame()
    for c in ctxs: c.synchronize()
    print(f"{parts} launches per frame on two streams, joined per frame: {(time.perf_counter() - t0) / N * 1e3:.4f} ms per frame (one launch: {one:.4f})")
