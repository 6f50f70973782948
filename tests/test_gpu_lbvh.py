"""The opt-in device-built tree (rt_scene_upload_lbvh: linear BVH over Morton-sorted primitives).
It is NOT the reference's tree, so it is checked against exhaustive intersection (the property the
reference's own BVHSelfTest checks, BVH.cpp:181-260) and against the traversal of the reference tree,
not against the goldens."""
import os

import numpy as np
import pytest

import raytracer_amd as rt
from conftest import CAM, DATA, build_scene
from oracle.bindings import Oracle

pytestmark = pytest.mark.gpu


def _rays(n, seed, lo=-1.5, hi=1.5):
    rng = np.random.default_rng(seed)
    org = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    return org, d


@pytest.mark.parametrize("mesh", ["teapot_normal.obj", "water.obj"])
def test_lbvh_equals_exhaustive_intersection(gpu_ctx, mesh):
    s = build_scene(rt.Scene(), mesh)
    orc = build_scene(Oracle(), mesh)
    ms = gpu_ctx.upload_lbvh(s)
    assert ms > 0
    org, d = _rays(40000, 3)
    ids, t = gpu_ctx.trace_closest(org, d)
    bid, bt = orc.brute_force(org, d)
    assert np.array_equal(t, bt)                      # the closest distance is the same number
    assert (ids == bid).mean() > 0.9999               # ids differ only on exact ties
    assert np.array_equal(ids >= 0, bid >= 0)
    # the same frame through the reference tree: same image
    p = rt.default_params(width=640, height=360)
    ids_l, t_l, _, _ = gpu_ctx.trace_primary(rt.look_at(CAM[:3], CAM[3:]), p)
    gpu_ctx.upload(s)
    ids_r, t_r, _, _ = gpu_ctx.trace_primary(rt.look_at(CAM[:3], CAM[3:]), p)
    # ids may differ where two triangles give the very same t (a ray through a shared edge, e.g. a wall's
    # diagonal): the distance image is identical
    assert np.array_equal(np.where(ids_l >= 0, t_l, 1e9), np.where(ids_r >= 0, t_r, 1e9))
    assert (ids_l != ids_r).mean() < 1e-3


def test_lbvh_mixed_primitives_small_and_empty_scenes(gpu_ctx):
    def fill(api, n_tri, n_sph):
        r2 = np.random.default_rng(21)
        for i in range(n_tri):
            A, C, B = r2.uniform(-1, 1, (3, 3))
            api.add_triangle(A, C, B, (0.3, 0.6, 0.9, 0.0), 1.5)
        for i in range(n_sph):
            api.add_sphere(r2.uniform(-0.8, 0.8, 3), float(r2.uniform(0.05, 0.4)), (1, 1, 1, 0.9), 1.5)
        api.add_plane((0.0, 1.0, 0.0), -1.0, (0.5, 0.5, 0.5, 0))
        api.build_bvh()
        return api
    org, d = _rays(20000, 4)
    for n_tri, n_sph in [(60, 0), (1, 0), (7, 0), (16, 0), (0, 0), (60, 30)]:
        s, orc = fill(rt.Scene(), n_tri, n_sph), fill(Oracle(), n_tri, n_sph)
        gpu_ctx.upload_lbvh(s)
        ids, t = gpu_ctx.trace_closest(org, d)
        sid, st_ = orc.shoot(org, d)      # the reference traversal (tree) + the plane side list
        if n_sph == 0:
            assert np.array_equal(t, st_), (n_tri, n_sph)
            assert (ids == sid).mean() > 0.999, (n_tri, n_sph)
        else:
            # Sphere::findIntersection's second root (Sphere.h:83) is not a geometric intersection, so which
            # spheres answer depends on which boxes a ray happens to enter: only near-agreement between trees
            assert (t == st_).mean() > 0.97
            assert ((ids >= 60) & (ids < 90)).any()


def test_lbvh_large_scene_and_render_path(gpu_ctx):
    """49 teapots: the device tree against the reference tree on a frame, in both precisions; then the
    whole render path (bounce queue) on a glass teapot with either tree."""
    s = rt.Scene()
    for i in range(7):
        for j in range(7):
            s.load_obj(os.path.join(DATA, "teapot_normal.obj"), off=(0.0, i * 1.5, j * 1.5))
    s.build_bvh()
    cam = rt.look_at((9.0, 4.5, 4.5), (0.0, 4.5, 4.5))
    out = {}
    for tree in ("lbvh", "reference"):
        ms = gpu_ctx.upload_lbvh(s) if tree == "lbvh" else gpu_ctx.upload(s)
        for prec in (rt.RT_F64, rt.RT_F32):
            out[tree, prec] = gpu_ctx.trace_primary(cam, rt.default_params(width=480, height=300, precision=prec))[:2]
    for prec in (rt.RT_F64, rt.RT_F32):
        il, tl = out["lbvh", prec]
        ir, tr = out["reference", prec]
        assert (il != ir).mean() < 1e-3, prec
        if prec == rt.RT_F64:
            assert np.array_equal(np.where(il >= 0, tl, 1e9), np.where(ir >= 0, tr, 1e9))
        else:   # float boxes are rounded, not widened: a hit within float rounding of a box face may be culled by one tree only
            assert (np.where(il >= 0, tl, 1e9) != np.where(ir >= 0, tr, 1e9)).mean() < 1e-3
    g = build_scene(rt.Scene(), "teapot_normal.obj", (1.0, 1.0, 1.0, 1.8), 1.5, (0, -0.7, 0))
    g.emit_photons(10000, seed=5)      # (no direct lighting in this renderer: a frame without photons is black)
    gpu_ctx.photons_upload(g.photons_get())
    p = rt.default_params(width=320, height=180, photon_num=10000, photon_use=50)
    gpu_ctx.upload(g)
    ref, _, _ = gpu_ctx.render(rt.look_at(CAM[:3], CAM[3:]), p)
    assert ref.max() > 0.2
    gpu_ctx.upload_lbvh(g)
    got, _, st = gpu_ctx.render(rt.look_at(CAM[:3], CAM[3:]), p)
    a, b = np.floor(np.clip(got, 0, 1) * 255), np.floor(np.clip(ref, 0, 1) * 255)
    # (glass: refracted rays that start on a box face see the two trees differently, see test_gpu_scene_api)
    assert (np.abs(a - b).max(-1) <= 1).mean() > 0.99
    assert st.secondary_rays > 0


def test_ten_million_triangles_two_trees_one_image(gpu_ctx):
    """The largest configuration at its full size: 10 000 teapots (9.92 M triangles, reference tree 29 deep) seen at
    7680x4320.  Two independently built trees -- the reference's on the host, the linear BVH on the device -- traced
    by the same kernel must find the same closest hit for every one of the 33 M camera rays (distance image
    identical; ids differ only where two triangles give the very same t); the float instantiation differs from the
    double one on silhouette pixels only."""
    side = 100
    s = rt.Scene()
    for i in range(side):
        for j in range(side):
            s.load_obj(os.path.join(DATA, "teapot_normal.obj"), (0.5, 0.5, 0.5, 0.0), 1.0, (0.0, i * 1.5, j * 1.5))
    s.build_bvh()
    assert s.size() == 992 * side * side and s.bvh_depth() >= 24
    c = side * 1.5 / 2
    cam = rt.look_at((side * 1.5 * 0.6, c, c), (0.0, c, c))
    w, h = 7680, 4320
    p = rt.default_params(width=w, height=h, shadow_rays=0)
    gpu_ctx.upload(s)
    ids_r, t_r, _, st = gpu_ctx.trace_primary(cam, p)
    assert st.primary_rays == w * h and (ids_r >= 0).sum() > 9000000
    p32 = rt.default_params(width=w, height=h, shadow_rays=0, precision=rt.RT_F32)
    ids_f, _, _, _ = gpu_ctx.trace_primary(cam, p32, want_t=False)
    assert (ids_f != ids_r).mean() < 1e-3                                   # float rounding moves silhouettes only
    ms = gpu_ctx.upload_lbvh(s)
    ids_l, t_l, _, _ = gpu_ctx.trace_primary(cam, p)
    assert np.array_equal(np.where(ids_l >= 0, t_l, 1e9), np.where(ids_r >= 0, t_r, 1e9))
    assert (ids_l != ids_r).mean() < 1e-4
    assert ms > 0 and ids_r.max() < s.size()

    # The same configuration through the full pixel path at 16 samples per pixel (aadepth 4: main.cpp:771-787 shoots
    # 16 camera rays per pixel of which 4 are distinct), 7680x4320, with a photon map so that the frame is not black:
    # 132.7 M traced camera rays per frame.  The frame of the device-built tree equals the frame of the reference's
    # tree (same photon map), and a pixel's 16-sample value lies within the range its neighbourhood spans at 1 sample.
    import ctypes
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")                    # the runtime the library itself uses

    def render_bgr8(params):
        d = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(h * w * 3)) == 0
        st_ = gpu_ctx.render_dev(cam, params, (0, h), None, d)
        out = np.zeros((h, w, 3), np.uint8)
        assert hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), d, ctypes.c_size_t(h * w * 3), 2) == 0
        assert hip.hipFree(d) == 0
        return out, st_
    # (photon_num only scales the radiance estimate, main.cpp:293: 160 instead of the emitted count lifts the frame
    # out of the lowest 8-bit levels, so that the comparison of the two frames means something)
    light = (side * 0.5, c, c)
    (n_surface, _), _ = gpu_ctx.emit_photon_maps(2_000_000, light_pos=light, seed=7)
    assert n_surface > 100_000
    frames = []
    for tree in ("device", "reference"):
        if tree == "reference":
            gpu_ctx.upload(s)
        p16 = rt.default_params(width=w, height=h, aadepth=4, photon_num=160, photon_use=50)
        frame, st16 = render_bgr8(p16)
        assert st16.primary_rays == w * h * 4 and st16.secondary_rays == 0
        assert st16.gather_queries > 4 * 9000000
        frames.append(frame)
    differ = (frames[0] != frames[1]).any(-1).mean()
    assert differ < 1e-3, differ
    assert frames[1].max() > 50 and (frames[1].sum(-1) > 0).mean() > 0.05     # a lit frame
    one = render_bgr8(rt.default_params(width=w, height=h, aadepth=1, photon_num=160, photon_use=50))[0].astype(np.int32)
    sixteen = frames[1].astype(np.int32)
    # camera rays of aadepth 4 leave pixel (x, y) between the centres of pixels (x - 1 .. x + 1, y - 1 .. y + 1)
    interior = (one[1:-1, 1:-1].sum(-1) > 0) & (one[:-2, 1:-1].sum(-1) > 0) & (one[2:, 1:-1].sum(-1) > 0) & (one[1:-1, :-2].sum(-1) > 0) & (one[1:-1, 2:].sum(-1) > 0)
    assert np.abs(sixteen[1:-1, 1:-1][interior].mean() - one[1:-1, 1:-1][interior].mean()) < 2.0
    gpu_ctx.photons_upload(np.zeros((0, 9)))
    gpu_ctx.upload(build_scene(rt.Scene()))                                  # leave a small scene behind
