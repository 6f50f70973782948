"""Frame assembly over processes through the C ABI (rt_comm_create / rt_gather_frame: RCCL bound at run time).
On a one-GPU box a communicator of one rank runs the whole call path -- RCCL loaded, communicator made, the rank's bands
put in their places; the exchange itself needs two GPUs and is skipped without them."""
import ctypes
import os
import sys

import numpy as np
import pytest

import raytracer_amd as rt
from conftest import CAM, ROOT, build_scene

pytestmark = pytest.mark.gpu

HIP = "/opt/rocm/lib/libamdhip64.so"      # the runtime the library itself uses


def _dmalloc(hip, n_bytes):
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n_bytes)) == 0
    return p


def _render_part(ctx, hip, cam, w, h, band, n, r):
    d_part = _dmalloc(hip, (h // n) * w * 4)
    p = rt.default_params(width=w, height=h, shadow_rays=0, band_rows=band, n_parts=n, part=r)
    ctx.trace_primary_dev(cam, p, (0, h // n), d_part, None, None, stats=False)      # enqueued, not waited for
    return d_part


def test_gather_frame_one_rank():
    hip = ctypes.CDLL(HIP)
    band, w, h = 8, 200, 96
    cam = rt.look_at(CAM[:3], CAM[3:])
    ctx = rt.Context(0)
    ctx.upload(build_scene(rt.Scene(), "teapot_normal.obj"))
    whole, _, _, _ = ctx.trace_primary(cam, rt.default_params(width=w, height=h, shadow_rays=0))
    comm = rt.Comm(ctx, rt.comm_unique_id(), 0, 1)
    d_part = _render_part(ctx, hip, cam, w, h, band, 1, 0)
    d_frame = _dmalloc(hip, h * w * 4)
    assert hip.hipMemset(d_frame, 0xFF, ctypes.c_size_t(h * w * 4)) == 0
    comm.gather_frame(d_part, band, h, w * 4, d_frame)
    ctx.synchronize()
    got = np.zeros(h * w, np.int32)
    assert hip.hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), d_frame, ctypes.c_size_t(h * w * 4), 2) == 0
    assert np.array_equal(got, whole.ravel())
    with pytest.raises(rt.RtError):
        comm.gather_frame(d_part, band, h + 1, w * 4, d_frame)          # does not tile: refused before anything is enqueued
    with pytest.raises(rt.RtError):
        comm.gather_frame(d_part, band, h, w * 4, None)                 # rank 0 without a frame
    comm.close()
    for p in (d_part, d_frame):
        assert hip.hipFree(p) == 0
    with pytest.raises(rt.RtError):
        rt.Comm(ctx, rt.comm_unique_id(), 2, 2)                         # rank outside the job


def _rank(rank, world, id_path, out_path):
    """One process, one GPU: trace this rank's bands, gather to rank 0."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import time
    import torch
    torch.cuda.init()
    import raytracer_amd as rtp
    from conftest import CAM as cam_v, build_scene
    hip = ctypes.CDLL(HIP)
    band, w, h = 8, 200, 8 * world * 6
    ctx = rtp.Context(rank)
    assert hip.hipSetDevice(rank) == 0
    ctx.upload(build_scene(rtp.Scene(), "teapot_normal.obj"))
    if rank == 0:
        with open(id_path + ".tmp", "wb") as f:
            f.write(rtp.comm_unique_id())
        os.replace(id_path + ".tmp", id_path)
    for _ in range(600):
        if os.path.exists(id_path):
            break
        time.sleep(0.1)
    comm = rtp.Comm(ctx, open(id_path, "rb").read(), rank, world)
    cam = rtp.look_at(cam_v[:3], cam_v[3:])
    d_part = _render_part(ctx, hip, cam, w, h, band, world, rank)
    d_frame = _dmalloc(hip, h * w * 4) if rank == 0 else None
    comm.gather_frame(d_part, band, h, w * 4, d_frame)
    ctx.synchronize()
    if rank == 0:
        got = np.zeros(h * w, np.int32)
        assert hip.hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), d_frame, ctypes.c_size_t(h * w * 4), 2) == 0
        whole, _, _, _ = ctx.trace_primary(cam, rtp.default_params(width=w, height=h, shadow_rays=0))
        np.save(out_path, np.stack([got, whole.ravel()]))
    comm.close()


def test_gather_frame_two_ranks(tmp_path):
    """Two processes on two GPUs: rank 1's bands travel over RCCL.  Needs a box with at least two GPUs."""
    from raytracer_amd import _lib
    if _lib.lib.rt_device_count() < 2:
        pytest.skip("one GPU on this box")
    import torch.multiprocessing as mp
    out = str(tmp_path / "frame.npy")
    mp.spawn(_rank, args=(2, str(tmp_path / "comm.id"), out), nprocs=2, join=True)
    got, whole = np.load(out)
    assert np.array_equal(got, whole)
