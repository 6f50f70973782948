"""The entry point that allocates nothing (rt_knn_estimate_dev_noalloc): requests and results in the caller's device
buffers, all working memory in a fixed array inside the library.  Its answers must be those of the entry point it is
a variant of, rt_knn_estimate_grouped (same kernels, same groups of 64 consecutive requests), on two calls in a row --
the second with fewer requests, so that whatever the first left in the array would show."""
import ctypes

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import _lib
from conftest import CAM, build_scene, golden

pytestmark = pytest.mark.gpu

LIMIT = -6


def test_no_alloc_gather_matches_grouped_entry_point():
    ctx = rt.Context(0)
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")                    # the runtime the library itself uses
    g = golden("photons_cornell.npz")
    ctx.photons_upload(g["photons"])
    s = build_scene(rt.Scene(), None)
    ctx.upload(s)
    # diffuse hit points of a frame in raster order: neighbouring requests, as the pixel path queues them
    w, h = 160, 120
    ids, t, _, _ = ctx.trace_primary(rt.look_at(CAM[:3], CAM[3:]), rt.default_params(width=w, height=h))
    cam = rt.look_at(CAM[:3], CAM[3:])
    from oracle.bindings import Oracle
    orc = build_scene(Oracle(), None)
    org, d = orc.primary_rays(w, h, CAM)
    hit = ids >= 0
    pos_all = org[hit] + d[hit] * (t[hit][:, None] + 1e-6)
    nrm_all = orc.obj_normal(ids[hit], pos_all)
    col_all = orc.obj_color(ids[hit], pos_all)[:, :3]
    assert len(pos_all) > 12000

    def dev(a):
        a = np.ascontiguousarray(a, np.float64)
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(max(a.nbytes, 8))) == 0
        assert hip.hipMemcpy(p, a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(a.nbytes), 1) == 0
        return p

    def free_bytes():
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    for k, n in ((1000, 12000), (1000, 3777), (50, 64), (50, 1)):          # second and later calls are smaller
        p = rt.default_params(photon_use=k)
        pos, nrm, col = pos_all[:n], nrm_all[:n], col_all[:n]
        want = ctx.knn_estimate_grouped(p, pos, nrm, col)
        d_pos, d_nrm, d_col = dev(pos), dev(nrm), dev(col)
        d_rgb = dev(np.full((n, 3), 7.0))                                   # stale values in the output must not survive
        ctx.synchronize()
        before = free_bytes()
        rc = _lib.lib.rt_knn_estimate_dev_noalloc(ctx.h, ctypes.byref(p), n, d_pos, d_nrm, d_col, d_rgb)
        assert rc == 0, _lib.lib.rt_last_error()
        during = free_bytes()                                               # (the kernels may still be running)
        ctx.synchronize()
        # the device's free memory is what it was: the call allocated nothing and freed nothing
        assert before == during == free_bytes(), (before, during, free_bytes())
        got = np.zeros((n, 3))
        assert hip.hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), d_rgb, ctypes.c_size_t(got.nbytes), 2) == 0
        for q in (d_pos, d_nrm, d_col, d_rgb):
            assert hip.hipFree(q) == 0
        assert want.max() > 0.002
        # same kernels over the same groups: the sums may run in another order only where a request went to the
        # one-request-per-wave kernel, whose candidate order depends on the bound its predecessor handed over
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, want.max()), (k, n, np.abs(got - want).max())
    # Two contexts of the one GPU call it back to back without waiting in between: both use the GPU's one array, so the
    # library orders the second behind the first, and each gets the answers it gets when called alone -- whatever the
    # other's requests left in the queue, the counters and the hand-over list.
    ctx2 = rt.Context(0)
    ctx2.photons_upload(g["photons"])
    ctx2.upload(s)
    p = rt.default_params(photon_use=1000)
    na, nb = 9000, 2500
    want_a = ctx.knn_estimate_grouped(p, pos_all[:na], nrm_all[:na], col_all[:na])
    want_b = ctx2.knn_estimate_grouped(p, pos_all[-nb:], nrm_all[-nb:], col_all[-nb:])
    bufs_a = [dev(x) for x in (pos_all[:na], nrm_all[:na], col_all[:na], np.full((na, 3), 7.0))]
    bufs_b = [dev(x) for x in (pos_all[-nb:], nrm_all[-nb:], col_all[-nb:], np.full((nb, 3), 7.0))]
    for rep in range(2):
        assert _lib.lib.rt_knn_estimate_dev_noalloc(ctx.h, ctypes.byref(p), na, *bufs_a) == 0, _lib.lib.rt_last_error()
        assert _lib.lib.rt_knn_estimate_dev_noalloc(ctx2.h, ctypes.byref(p), nb, *bufs_b) == 0, _lib.lib.rt_last_error()
    ctx.synchronize(); ctx2.synchronize()
    for bufs, want, n in ((bufs_a, want_a, na), (bufs_b, want_b, nb)):
        got = np.zeros((n, 3))
        assert hip.hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), bufs[3], ctypes.c_size_t(got.nbytes), 2) == 0
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, want.max()), np.abs(got - want).max()
        for q in bufs:
            assert hip.hipFree(q) == 0
    del ctx2
    # beyond the fixed working memory: refused, nothing launched
    p = rt.default_params()
    d_any = dev(np.zeros(8))
    rc = _lib.lib.rt_knn_estimate_dev_noalloc(ctx.h, ctypes.byref(p), 65537, d_any, d_any, d_any, d_any)
    assert rc == LIMIT and b"65537" in _lib.lib.rt_last_error()
    assert _lib.lib.rt_knn_estimate_dev_noalloc(ctx.h, ctypes.byref(p), 0, d_any, d_any, d_any, d_any) == 0
    assert hip.hipFree(d_any) == 0
