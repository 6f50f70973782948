"""Error behaviour and empty inputs of the C-ABI on a GPU: every misuse returns a negative status with a
message (RtError), nothing faults, and the context stays usable afterwards."""
import ctypes

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import _lib
from conftest import CAM, build_scene

pytestmark = pytest.mark.gpu

INVALID, STATE, LIMIT = -1, -4, -6


def code_of(fn, *a, **kw):
    with pytest.raises(rt.RtError) as e:
        fn(*a, **kw)
    assert "status" in str(e.value) and len(str(e.value)) > 20      # a message comes with the code
    return e.value.code


def raw_render(ctx, cam, p, r0=0, r1=32):
    """rt_render_rows with buffers large enough for any 32-wide case: the status as returned."""
    rgb = np.zeros((64, 64, 3))
    bgr = np.zeros((64, 64, 3), np.uint8)
    return _lib.lib.rt_render_rows(ctx.h, ctypes.byref(cam), ctypes.byref(p), r0, r1, rgb.ctypes.data_as(_lib.P(_lib.c_d)),
                                   bgr.ctypes.data_as(_lib.P(_lib.c_u8)), None)


@pytest.fixture(scope="module")
def teapot():
    ctx = rt.Context(0)
    ctx.upload(build_scene(rt.Scene()))
    return ctx


def test_calls_before_upload_report_state():
    ctx = rt.Context(0)
    cam = rt.look_at(CAM[:3], CAM[3:])
    p = rt.default_params(width=16, height=16)
    assert code_of(ctx.trace_closest, np.zeros((1, 3)), np.ones((1, 3))) == STATE
    assert code_of(ctx.trace_primary, cam, p) == STATE
    assert code_of(ctx.render, cam, p) == STATE
    assert code_of(ctx.emit_photon_maps, 100) == STATE
    s = rt.Scene()
    s.add_cornell()
    assert code_of(ctx.upload, s) == STATE                           # the tree has not been built
    assert code_of(s.emit_photons, 10) == STATE


def test_bad_arguments(teapot):
    ctx = teapot
    cam = rt.look_at(CAM[:3], CAM[3:])
    good = rt.default_params(width=32, height=32)
    assert code_of(ctx.trace_closest, np.zeros((1, 3)), np.ones((1, 3)), precision=7) == INVALID
    for kw in (dict(width=0), dict(height=-4), dict(precision=5), dict(struct_size=8), dict(band_rows=8, n_parts=3, part=0),
               dict(band_rows=8, n_parts=2, part=2)):
        assert raw_render(ctx, cam, rt.default_params(**{**dict(width=32, height=32), **kw})) == INVALID, kw
    assert raw_render(ctx, cam, good, 0, 33) == INVALID
    assert raw_render(ctx, cam, good, 5, 5) == INVALID
    assert raw_render(ctx, cam, good, -1, 8) == INVALID
    assert raw_render(ctx, cam, good, 0, 32) == 0
    assert code_of(ctx.render, cam, rt.default_params(width=32, height=32, aadepth=0)) == INVALID
    assert code_of(ctx.render, cam, rt.default_params(width=32, height=32, dof_samples=0)) == INVALID
    assert code_of(ctx.render, cam, rt.default_params(width=32, height=32, tracing_depth=0)) == INVALID
    assert code_of(ctx.render, cam, rt.default_params(width=32, height=32, photon_use=100000)) == LIMIT
    assert code_of(ctx.trace_primary, cam, rt.default_params(width=32, height=32, aadepth=2)) == INVALID
    assert code_of(ctx.knn_radius, 0, np.zeros((4, 3))) == LIMIT
    assert code_of(ctx.knn_radius, 1 << 20, np.zeros((4, 3))) == LIMIT
    assert code_of(ctx.knn_radius, 8, np.zeros((4, 3)), which=2) == INVALID
    null = ctypes.c_void_p()
    assert _lib.lib.rt_scene_upload(null, null) == INVALID
    assert _lib.lib.rt_context_synchronize(null) == INVALID
    assert _lib.lib.rt_context_create(10 ** 6, ctypes.byref(ctypes.c_void_p())) == INVALID
    assert b"out of range" in _lib.lib.rt_last_error()
    # the context still works
    ids, _, _, _ = ctx.trace_primary(cam, rt.default_params(width=32, height=32, shadow_rays=0))
    assert (ids >= 0).any()


def test_empty_inputs(teapot):
    ctx = teapot
    ids, dist = ctx.trace_closest(np.zeros((0, 3)), np.zeros((0, 3)))
    assert ids.shape == (0,) and dist.shape == (0,)
    ctx.photons_upload(np.zeros((0, 9)), 0)
    ctx.photons_upload_dev_build(np.zeros((0, 9)), 1)
    cnt, r2 = ctx.knn_radius(8, np.zeros((5, 3)))
    assert (cnt == 0).all()
    idx, cnt = ctx.knn_query(8, np.zeros((0, 3)))
    assert idx.shape == (0, 8)
    p = rt.default_params(width=32, height=32)
    rgb, cnt = ctx.knn_estimate(p, np.zeros((3, 3)), np.tile([0., 0, 1], (3, 1)), np.ones((3, 3)))
    assert (rgb == 0).all() and (cnt == 0).all()
    assert (ctx.knn_estimate_grouped(p, np.zeros((3, 3)), np.tile([0., 0, 1], (3, 1)), np.ones((3, 3))) == 0).all()
    cam = rt.look_at(CAM[:3], CAM[3:])
    rgb, bgr, _ = ctx.render(cam, p)                                   # photon map asked for, none stored: black, as in the reference
    assert (bgr == 0).all()


def test_volume_map_without_surface_map(teapot):
    """A stored volume map beside an empty surface map: diffuse hits gather nothing and nothing faults."""
    ctx = teapot
    rng = np.random.default_rng(3)
    ph = np.c_[rng.uniform(-0.5, 0.5, (500, 3)), np.tile([0., 0, 1], (500, 1)), np.full((500, 3), 0.5)]
    ctx.photons_upload(np.zeros((0, 9)), 0)
    ctx.photons_upload(ph, 1)
    cam = rt.look_at(CAM[:3], CAM[3:])
    rgb, bgr, st = ctx.render(cam, rt.default_params(width=64, height=64))
    assert (bgr == 0).all() and np.isfinite(rgb).all()
    ctx.photons_upload(np.zeros((0, 9)), 1)


def test_queue_overflow_is_reported(monkeypatch):
    """A bounce queue too small for the frame (forced through RT_SHARD_CAP): the call that waits reports it itself, the
    call that does not wait leaves it to the next synchronize, and the report is made once."""
    ctx = rt.Context(0)
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")                                # the runtime the library itself uses
    d_rgb = ctypes.c_void_p()
    n_bytes = 256 * 256 * 3 * 8
    assert hip.hipMalloc(ctypes.byref(d_rgb), ctypes.c_size_t(n_bytes)) == 0
    ctx.upload(build_scene(rt.Scene(), rgbs=(1.0, 1.0, 1.0, 1.8), ior=1.5))           # glass teapot: every hit spawns two rays
    cam = rt.look_at(CAM[:3], CAM[3:])
    p = rt.default_params(width=256, height=256, use_photon_map=0)
    ctx.render(cam, p)                                                               # fits: no error
    ctx.synchronize()
    monkeypatch.setenv("RT_SHARD_CAP", "64")
    assert code_of(ctx.render, cam, p) == LIMIT
    ctx.synchronize()                                                                # already reported
    ctx.render_dev(cam, p, (0, 256), d_rgb, None, stats=False)                       # does not wait, returns OK
    assert code_of(ctx.synchronize) == LIMIT
    ctx.synchronize()                                                                # reported once
    monkeypatch.delenv("RT_SHARD_CAP")
    ref, _, _ = ctx.render(cam, p)
    ctx.render_dev(cam, p, (0, 256), d_rgb, None, stats=False)
    ctx.synchronize()
    out = np.zeros((256, 256, 3))
    assert hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), d_rgb, ctypes.c_size_t(n_bytes), 2) == 0      # device to host
    assert hip.hipFree(d_rgb) == 0
    assert np.abs(out - ref).max() < 1e-12
