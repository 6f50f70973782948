"""GPU parity of the closest-hit path (through the C ABI) against the CPU oracle and the golden
vectors of the compiled reference.  f64 instantiation: ids and t bit-exact."""
import hashlib
import os

import numpy as np
import pytest

import raytracer_amd as rt
from conftest import CAM, DATA, build_scene, golden, golden_hashes
from oracle.bindings import Oracle, id_map_hash

pytestmark = pytest.mark.gpu


def eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


@pytest.fixture(scope="module")
def scenes(gpu_ctx):
    out = {}
    for tag in ["teapot_normal", "water"]:
        out[tag] = build_scene(rt.Scene(), tag + ".obj")
    return out


@pytest.mark.parametrize("tag", ["teapot_normal", "water"])
def test_batch_rays_bit_exact(gpu_ctx, scenes, tag):
    gpu_ctx.upload(scenes[tag])
    g = golden(f"rays_{tag}.npz")
    ids, t = gpu_ctx.trace_closest(g["org"], g["dir"])
    assert eq(ids, g["ids"]) and eq(t, g["t"])
    ids, t = gpu_ctx.trace_closest(g["org"], g["dir"], g["max_dist"])
    assert eq(ids, g["ids_bounded"]) and eq(t, g["t_bounded"])
    # axis-parallel rays: zero direction components -> inf / NaN paths of the slab test
    ids, t = gpu_ctx.trace_closest(g["org_axis"], g["dir_axis"])
    assert eq(ids, g["ids_axis"]) and eq(t, g["t_axis"])
    for sz in ["64x64", "96x54", "40x72"]:
        h = golden(f"hits_{tag}_{sz}.npz")
        org = np.repeat(h["org"], h["dir"].shape[0], axis=0)
        ids, t = gpu_ctx.trace_closest(org, h["dir"])
        assert eq(ids, h["ids"]) and eq(t, h["t"])


@pytest.mark.parametrize("env", [{"RT_TRAV_MODE": "retest"}, {"RT_TRAV_MODE": "plain", "RT_NO_NODE_LDS": "1"}, {"RT_TRAV_MODE": "plain", "RT_BLOCK_WAVES": "2"}])
def test_every_stack_form_bit_exact(gpu_ctx, scenes, monkeypatch, env):
    """The library picks the form of the traversal stack and whether the node array is staged in LDS from the tree's
    depth and size; each form, forced here, gives the reference's hits on the golden rays (batch entry point, bounded and
    axis-parallel rays included) and on a golden frame (pixel entry point)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    gpu_ctx.upload(scenes["teapot_normal"])
    g = golden("rays_teapot_normal.npz")
    for prec in (rt.RT_F64,):
        ids, t = gpu_ctx.trace_closest(g["org"], g["dir"], precision=prec)
        assert eq(ids, g["ids"]) and eq(t, g["t"])
        ids, t = gpu_ctx.trace_closest(g["org"], g["dir"], g["max_dist"], precision=prec)
        assert eq(ids, g["ids_bounded"]) and eq(t, g["t_bounded"])
        ids, t = gpu_ctx.trace_closest(g["org_axis"], g["dir_axis"], precision=prec)
        assert eq(ids, g["ids_axis"]) and eq(t, g["t_axis"])
    h = golden("hits_teapot_normal_96x54.npz")
    ids, t, _, _ = gpu_ctx.trace_primary(rt.look_at(CAM[:3], CAM[3:]), rt.default_params(width=96, height=54))
    assert eq(ids.ravel(), h["ids"].ravel()) and eq(np.where(ids >= 0, t, 1e9).ravel(), h["t"].ravel())
    # the float walk in the same form equals the float walk in the form the library would pick
    ids32, t32 = gpu_ctx.trace_closest(g["org"], g["dir"], precision=rt.RT_F32)
    for k in env:
        monkeypatch.delenv(k)
    ref32, reft32 = gpu_ctx.trace_closest(g["org"], g["dir"], precision=rt.RT_F32)
    assert eq(ids32, ref32) and eq(t32, reft32)


@pytest.mark.parametrize("tag", ["teapot_normal", "water"])
def test_primary_1920x1080_hash(gpu_ctx, scenes, tag):
    """The full-size id map: hash, hit counts and sha256 of the t image equal the compiled reference's."""
    gpu_ctx.upload(scenes[tag])
    p = rt.default_params(width=1920, height=1080)
    ids, t, _, st = gpu_ctx.trace_primary(rt.look_at(CAM[:3], CAM[3:]), p)
    g = golden_hashes()[tag]
    assert id_map_hash(ids) == g["hash"]
    assert int((ids >= 0).sum()) == g["hits"]
    t_ref_form = np.where(ids >= 0, t, 1e9)
    assert hashlib.sha256(t_ref_form.tobytes()).hexdigest() == g["t_sha256"]
    assert st.primary_rays == 1920 * 1080


def test_primary_small_sizes_and_row_bands(gpu_ctx, scenes):
    gpu_ctx.upload(scenes["teapot_normal"])
    cam = rt.look_at(CAM[:3], CAM[3:])
    for (w, h) in [(64, 64), (96, 54), (40, 72)]:
        g = golden(f"hits_teapot_normal_{w}x{h}.npz")
        p = rt.default_params(width=w, height=h)
        ids, t, _, _ = gpu_ctx.trace_primary(cam, p)
        assert eq(ids, g["ids"])
        assert eq(np.where(ids >= 0, t, 1e9), g["t"])
        # ragged row band (multi-GPU tiling unit): rows [5, 23)
        ids_b, t_b, _, _ = gpu_ctx.trace_primary(cam, p, rows=(5, 23))
        assert eq(ids_b, g["ids"].reshape(h, w)[5:23].ravel())


def test_primary_ragged_frames_match_oracle(gpu_ctx, scenes):
    """Frames whose sides are not multiples of the 8x8 tile or of the 16x8-tile super-blocks the waves walk:
    every pixel is traced exactly once (ids and t equal to the restatement's, none left at the fill value)."""
    gpu_ctx.upload(scenes["teapot_normal"])
    orc = build_scene(Oracle(), "teapot_normal.obj")
    cam = rt.look_at(CAM[:3], CAM[3:])
    for (w, h) in [(1, 1), (7, 3), (50, 37), (131, 67), (257, 130)]:
        p = rt.default_params(width=w, height=h)
        ids, t, _, st = gpu_ctx.trace_primary(cam, p)
        org, d = orc.primary_rays(w, h, CAM)
        oid, ot = orc.shoot(org, d)
        assert eq(ids, oid), (w, h)
        assert eq(np.where(ids >= 0, t, 1e9), np.where(oid >= 0, ot, 1e9)), (w, h)
        assert st.primary_rays == w * h
        if w >= 50:                                       # the pixel path walks the same tiles (glass: two bounces deep)
            rgb, _, _ = gpu_ctx.render(cam, rt.default_params(width=w, height=h, use_photon_map=0))
            assert rgb.shape == (h, w, 3) and np.isfinite(rgb).all()


@pytest.mark.parametrize("n_parts", [2, 4, 8])
def test_primary_row_bands_of_all_parts_make_the_frame(gpu_ctx, scenes, n_parts):
    """The multi-GPU split of the primary-ray path: every part traces its bands of 8 rows into a compact buffer;
    de-interleaved, the parts are the frame one GPU traces alone (ids, t and shadow ids)."""
    from raytracer_amd import tiling
    gpu_ctx.upload(scenes["teapot_normal"])
    cam = rt.look_at(CAM[:3], CAM[3:])
    w, h = 200, tiling.BAND * n_parts * 3
    whole = gpu_ctx.trace_primary(cam, rt.default_params(width=w, height=h, shadow_rays=1))
    parts = [gpu_ctx.trace_primary(cam, rt.default_params(width=w, height=h, shadow_rays=1, band_rows=tiling.BAND, n_parts=n_parts, part=r))
             for r in range(n_parts)]
    for k in range(3):
        got = tiling.deinterleave([pt[k].reshape(h // n_parts, w) for pt in parts])
        assert eq(got.ravel(), whole[k]), (n_parts, k)
    assert sum(pt[3].shadow_rays for pt in parts) == whole[3].shadow_rays


def test_gather_rows_assembles_the_frame_on_the_device(scenes):
    """rt_gather_rows (several GPUs in one process): three contexts -- on a one-GPU box all on device 0 -- trace their
    bands into compact device buffers; the strided device copies put every band in its place in the frame."""
    import ctypes
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")                    # the runtime the library itself uses
    n_parts, band = 3, 8
    w, h = 200, band * n_parts * 4
    cam = rt.look_at(CAM[:3], CAM[3:])
    ctxs = [rt.Context(0) for _ in range(n_parts)]
    for c in ctxs:
        c.upload(scenes["teapot_normal"])
    whole, _, _, _ = ctxs[0].trace_primary(cam, rt.default_params(width=w, height=h, shadow_rays=0))

    def dmalloc(n_bytes):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n_bytes)) == 0
        return p
    part_bytes = (h // n_parts) * w * 4
    d_parts = [dmalloc(part_bytes) for _ in range(n_parts)]
    d_frame = dmalloc(h * w * 4)
    assert hip.hipMemset(d_frame, 0xFF, ctypes.c_size_t(h * w * 4)) == 0
    for r, c in enumerate(ctxs):
        p = rt.default_params(width=w, height=h, shadow_rays=0, band_rows=band, n_parts=n_parts, part=r)
        c.trace_primary_dev(cam, p, (0, h // n_parts), d_parts[r], None, None, stats=False)      # enqueued, not waited for
    rt.gather_rows(ctxs, d_parts, band, h, w * 4, d_frame)
    got = np.zeros(h * w, np.int32)
    assert hip.hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), d_frame, ctypes.c_size_t(h * w * 4), 2) == 0
    for p in d_parts + [d_frame]:
        assert hip.hipFree(p) == 0
    assert eq(got, whole)
    with pytest.raises(rt.RtError):
        rt.gather_rows(ctxs, d_parts, band, h + 1, w * 4, d_frame)          # does not tile: refused before any copy


def test_gather_rows_across_two_devices(scenes):
    """The same assembly with the parts on two DIFFERENT GPUs (peer copies over xGMI, or the staged path where the
    devices cannot reach each other).  Needs a box with at least two GPUs."""
    import ctypes
    from raytracer_amd import _lib
    if _lib.lib.rt_device_count() < 2:
        pytest.skip("one GPU on this box")
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
    n_parts, band = 2, 8
    w, h = 200, band * n_parts * 4
    cam = rt.look_at(CAM[:3], CAM[3:])
    ctxs = [rt.Context(0), rt.Context(1)]
    for c in ctxs:
        c.upload(scenes["teapot_normal"])
    whole, _, _, _ = ctxs[0].trace_primary(cam, rt.default_params(width=w, height=h, shadow_rays=0))

    def dmalloc(dev, n_bytes):
        p = ctypes.c_void_p()
        assert hip.hipSetDevice(dev) == 0 and hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n_bytes)) == 0
        return p
    d_parts = [dmalloc(r, (h // n_parts) * w * 4) for r in range(n_parts)]
    d_frame = dmalloc(0, h * w * 4)
    assert hip.hipMemset(d_frame, 0xFF, ctypes.c_size_t(h * w * 4)) == 0
    for r, c in enumerate(ctxs):
        p = rt.default_params(width=w, height=h, shadow_rays=0, band_rows=band, n_parts=n_parts, part=r)
        c.trace_primary_dev(cam, p, (0, h // n_parts), d_parts[r], None, None, stats=False)
    rt.gather_rows(ctxs, d_parts, band, h, w * 4, d_frame)
    got = np.zeros(h * w, np.int32)
    assert hip.hipSetDevice(0) == 0
    assert hip.hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), d_frame, ctypes.c_size_t(h * w * 4), 2) == 0
    assert eq(got, whole)


def test_shadow_rays_match_oracle(gpu_ctx, scenes):
    """Occlusion rays (extension) go through the same Shoot semantics with a bounded range."""
    gpu_ctx.upload(scenes["teapot_normal"])
    orc = build_scene(Oracle(), "teapot_normal.obj")
    w, h = 160, 90
    light = np.array([0.1, 0.9, 0.0])
    p = rt.default_params(width=w, height=h, shadow_rays=1, light_pos=light)
    ids, t, sh, st = gpu_ctx.trace_primary(rt.look_at(CAM[:3], CAM[3:]), p)
    org, d = orc.primary_rays(w, h, CAM)
    oid, ot = orc.shoot(org, d)
    assert eq(ids, oid)
    hit = oid >= 0
    P = org[hit] + d[hit] * ot[hit][:, None]
    Lv = light - P
    ln = np.sqrt(Lv[:, 0] * Lv[:, 0] + Lv[:, 1] * Lv[:, 1] + Lv[:, 2] * Lv[:, 2])
    L = Lv / ln[:, None]
    so = P + L * 1e-4
    sid, _ = orc.shoot(so, L, ln - 1e-4)
    exp = np.full(w * h, -2, np.int32)
    exp[hit] = sid
    assert eq(sh, exp)
    assert st.shadow_rays == int(hit.sum())


@pytest.mark.parametrize("scene", ["teapot_normal", "water"])
@pytest.mark.parametrize("precision", [rt.RT_F64, rt.RT_F32])
def test_folded_shadow_bit_is_the_occlusion_of_the_shadow_ids(gpu_ctx, scenes, scene, precision):
    """shadow_rays = 2 without a shadow-id buffer: one word per pixel, bit 30 set where the shadow ray is occluded.  That
    query stops at the first occluder and walks the segment from the light; its answer is `shadow id >= 0` of the full
    walk from the surface except where rounding decides (a ray grazing an edge), and exactly the CPU oracle's Shoot on
    the same query (origin the light, direction -L, range |light - P| - 1e-4)."""
    import ctypes
    gpu_ctx.upload(scenes[scene])
    w, h = 640, 360
    cam = rt.look_at(CAM[:3], CAM[3:])
    light = np.array([0.1, 0.9, 0.0])
    ids, _, sh, st1 = gpu_ctx.trace_primary(cam, rt.default_params(width=w, height=h, shadow_rays=1, light_pos=light, precision=precision))
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")                                # the runtime the library itself uses
    d_ids = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d_ids), ctypes.c_size_t(w * h * 4)) == 0
    st2 = gpu_ctx.trace_primary_dev(cam, rt.default_params(width=w, height=h, shadow_rays=2, light_pos=light, precision=precision), (0, h), d_ids, None, None)
    folded = np.zeros(w * h, np.int32)
    assert hip.hipMemcpy(folded.ctypes.data_as(ctypes.c_void_p), d_ids, ctypes.c_size_t(w * h * 4), 2) == 0      # device to host
    assert hip.hipFree(d_ids) == 0
    bit = 1 << 30
    assert eq(np.where(folded >= 0, folded & ~bit, folded), ids)
    occluded = (folded >= 0) & ((folded & bit) != 0)
    differ = float((occluded != (sh >= 0)).mean())
    assert differ < 1e-4 and 0.02 < occluded.mean() < 0.9, differ
    assert st2.shadow_rays == st1.shadow_rays
    # the query itself, against the oracle (the float instantiation against the single-precision restatement)
    orc = build_scene(Oracle(), scene + ".obj")
    org, d = orc.primary_rays(w, h, CAM)
    oid, ot = orc.shoot(org, d)
    if precision == rt.RT_F32:
        oid, ot = orc.shoot_f32(org, d)
    assert eq(oid, ids)
    hit = oid >= 0
    P = org[hit] + d[hit] * ot[hit][:, None]
    Lv = light - P
    ln = np.sqrt(Lv[:, 0] * Lv[:, 0] + Lv[:, 1] * Lv[:, 1] + Lv[:, 2] * Lv[:, 2])
    L = Lv / ln[:, None]
    from_light = np.broadcast_to(light, L.shape).copy()
    shoot = orc.shoot_f32 if precision == rt.RT_F32 else orc.shoot
    sid, _ = shoot(from_light, -L, ln - 1e-4)
    exp = np.zeros(w * h, bool)
    exp[hit] = sid >= 0
    assert eq(occluded, exp)


def test_mixed_primitives_and_planes(gpu_ctx):
    """Spheres (with the reference's root quirks) inside the BVH, a plane in the side list."""
    def fill(api):
        r2 = np.random.default_rng(21)
        for i in range(60):
            A, C, B = r2.uniform(-1, 1, (3, 3))
            api.add_triangle(A, C, B, (0.3, 0.6, 0.9, 0.0), 1.5)
        for i in range(30):
            api.add_sphere(r2.uniform(-0.8, 0.8, 3), float(r2.uniform(0.05, 0.4)), (1, 1, 1, 0.9), 1.5)
        api.add_plane((0.0, 1.0, 0.0), -1.0, (0.5, 0.5, 0.5, 0))
        api.build_bvh()
        return api
    s = fill(rt.Scene()); orc = fill(Oracle())
    gpu_ctx.upload(s)
    rng = np.random.default_rng(4)
    org = rng.uniform(-1.5, 1.5, (20000, 3)); d = rng.normal(size=(20000, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    ids, t = gpu_ctx.trace_closest(org, d)
    oid, ot = orc.shoot(org, d)
    assert eq(ids, oid) and eq(t, ot)
    assert (ids == 90).any() and ((ids >= 60) & (ids < 90)).any()


def test_empty_and_tiny_scenes(gpu_ctx):
    s = rt.Scene(); s.build_bvh()
    gpu_ctx.upload(s)
    ids, t = gpu_ctx.trace_closest(np.zeros((5, 3)), np.ones((5, 3)))
    assert (ids == -1).all() and (t == 1e9).all()
    s = build_scene(rt.Scene(), None)    # Cornell box only: root is a leaf
    orc = build_scene(Oracle(), None)
    gpu_ctx.upload(s)
    rng = np.random.default_rng(8)
    org = rng.uniform(-1, 1, (5000, 3)); d = rng.uniform(-1, 1, (5000, 3))
    d[::50] = np.nan   # NaN direction -> miss (BVH.cpp:110-115)
    ids, t = gpu_ctx.trace_closest(org, d)
    oid, ot = orc.shoot(org, d)
    assert eq(ids, oid) and eq(t, ot)


def test_bvh_equals_brute_force_on_gpu(gpu_ctx, scenes):
    """BVHSelfTest's property (BVH.cpp:220-246) with the GPU traversal on the BVH side."""
    gpu_ctx.upload(scenes["water"])
    orc = build_scene(Oracle(), "water.obj")
    rng = np.random.default_rng(6)
    org = rng.uniform(-1, 1, (4000, 3)); d = rng.uniform(-1, 1, (4000, 3))
    ids, t = gpu_ctx.trace_closest(org, d)
    bids, bt = orc.brute_force(org, d)
    assert eq(t, bt)
    assert (ids == bids).mean() > 0.999


def test_instanced_grid_deep_tree(gpu_ctx):
    """49 teapots on a grid (48 608 triangles, a tree about 17 levels deep, node array too large
    for LDS staging): primary and random rays bit-exact against the oracle, both entry points."""
    def build(s):
        for i in range(7):
            for j in range(7):
                s.load_obj(os.path.join(DATA, "teapot_normal.obj"), off=(0.0, i * 1.5, j * 1.5))
        s.build_bvh()
        return s
    scene, orc = build(rt.Scene()), build(Oracle())
    assert scene.bvh_depth() >= 14
    gpu_ctx.upload(scene)
    cam_v = np.array([9.0, 4.5, 4.5, 0.0, 4.5, 4.5])
    w, h = 320, 200
    ids, t, _, _ = gpu_ctx.trace_primary(rt.look_at(cam_v[:3], cam_v[3:]), rt.default_params(width=w, height=h))
    org, d = orc.primary_rays(w, h, cam_v)
    oid, ot = orc.shoot(org, d)
    assert (oid >= 0).mean() > 0.2
    assert np.array_equal(ids, oid)
    assert np.array_equal(np.where(ids >= 0, t, 1e9), ot)
    rng = np.random.default_rng(5)
    n = 50_000
    o = rng.uniform(-1, 10, (n, 3))
    dd = rng.normal(size=(n, 3))
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    gid, gt = gpu_ctx.trace_closest(o, dd)
    rid, rdist = orc.shoot(o, dd)
    assert np.array_equal(gid, rid)
    assert np.array_equal(gt, rdist)


def _mixed_scene(api):
    r2 = np.random.default_rng(21)
    for i in range(60):
        A, C, B = r2.uniform(-1, 1, (3, 3))
        api.add_triangle(A, C, B, (0.3, 0.6, 0.9, 0.0), 1.5)
    for i in range(30):
        api.add_sphere(r2.uniform(-0.8, 0.8, 3), float(r2.uniform(0.05, 0.4)), (1, 1, 1, 0.9), 1.5)
    api.add_plane((0.0, 1.0, 0.0), -1.0, (0.5, 0.5, 0.5, 0))
    api.build_bvh()
    return api


def test_f32_instantiation_bit_exact(gpu_ctx, scenes):
    """The float instantiation of the kernels against the oracle's single-precision restatement (same
    tree, scene data and rays rounded to float, every operation in float): ids and t bit for bit --
    primary frames, random / bounded / axis-parallel rays (NaN paths), spheres and planes, a deep tree."""
    for tag in ["teapot_normal", "water"]:
        gpu_ctx.upload(scenes[tag])
        orc = build_scene(Oracle(), tag + ".obj")
        w, h = 640, 360
        ids, t, _, _ = gpu_ctx.trace_primary(rt.look_at(CAM[:3], CAM[3:]), rt.default_params(width=w, height=h, precision=rt.RT_F32))
        org, d = orc.primary_rays(w, h, CAM)
        oid, ot = orc.shoot_f32(org, d)
        assert eq(ids, oid), tag
        assert eq(np.where(ids >= 0, t, 1e9), ot), tag
        g = golden(f"rays_{tag}.npz")
        for o_, d_, m_ in [(g["org"], g["dir"], 1e9), (g["org"], g["dir"], g["max_dist"]), (g["org_axis"], g["dir_axis"], 1e9)]:
            gid, gt = gpu_ctx.trace_closest(o_, d_, m_, precision=rt.RT_F32)
            rid, rdist = orc.shoot_f32(o_, d_, m_)
            assert eq(gid, rid), tag
            # misses keep the caller's double range untouched on both sides
            assert eq(gt, rdist), tag
    s, orc = _mixed_scene(rt.Scene()), _mixed_scene(Oracle())
    gpu_ctx.upload(s)
    rng = np.random.default_rng(4)
    org = rng.uniform(-1.5, 1.5, (20000, 3)); d = rng.normal(size=(20000, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    gid, gt = gpu_ctx.trace_closest(org, d, precision=rt.RT_F32)
    rid, rdist = orc.shoot_f32(org, d)
    assert eq(gid, rid) and eq(gt, rdist)

    def grid(api):
        for i in range(7):
            for j in range(7):
                api.load_obj(os.path.join(DATA, "teapot_normal.obj"), off=(0.0, i * 1.5, j * 1.5))
        api.build_bvh()
        return api
    s, orc = grid(rt.Scene()), grid(Oracle())
    gpu_ctx.upload(s)
    cam_v = np.array([9.0, 4.5, 4.5, 0.0, 4.5, 4.5])
    ids, t, _, _ = gpu_ctx.trace_primary(rt.look_at(cam_v[:3], cam_v[3:]), rt.default_params(width=320, height=200, precision=rt.RT_F32))
    org, d = orc.primary_rays(320, 200, cam_v)
    oid, ot = orc.shoot_f32(org, d)
    assert eq(ids, oid) and eq(np.where(ids >= 0, t, 1e9), ot)


@pytest.mark.parametrize("tag", ["teapot1", "test"])
def test_meshes_without_normals(gpu_ctx, tag):
    """teapot1.obj (`f v/vt` faces: the mesh the second BASELINE configuration names) and test.obj through
    rt_scene_load_obj: tree, ids and t bit for bit against what the compiled reference traced on the same triangles
    (built by the harness's reader, since the reference's own crashes on faces without a normal index, IO.h:93)."""
    g = golden(f"mesh_{tag}.npz")
    s = build_scene(rt.Scene(), tag + ".obj")
    rec, boxes = s.bvh_dump()
    assert eq(rec, g["rec"]) and eq(boxes, g["boxes"])
    gpu_ctx.upload(s)
    cam = rt.look_at(CAM[:3], CAM[3:])
    ids, t, _, _ = gpu_ctx.trace_primary(cam, rt.default_params(width=64, height=64))
    assert eq(ids, g["ids64"]) and eq(np.where(ids >= 0, t, 1e9), g["t64"])
    ids, t = gpu_ctx.trace_closest(g["org"], g["dir"])
    assert eq(ids, g["ids"]) and eq(t, g["t"])
    ids, t, _, st = gpu_ctx.trace_primary(cam, rt.default_params(width=1920, height=1080))
    h = golden_hashes("idmap_hashes_no_normals.txt")[tag]
    assert id_map_hash(ids) == h["hash"] and int((ids >= 0).sum()) == h["hits"]
    assert int(((ids >= 0) & (ids < s.size() - 10)).sum()) == h["mesh_hits"]
    assert hashlib.sha256(np.where(ids >= 0, t, 1e9).tobytes()).hexdigest() == h["t_sha256"]
    # flat-normal shading of the same mesh: the frame the reference's getColorAt produced
    s.emit_photons(3000, seed=12345)
    gpu_ctx.photons_upload(s.photons_get())
    rgb, _, _ = gpu_ctx.render(cam, rt.default_params(width=64, height=64))
    a = np.floor(np.clip(rgb, 0, 1) * 255).astype(np.int32); b = np.floor(np.clip(g["rgb64"], 0, 1) * 255).astype(np.int32)
    assert (np.abs(a - b).max(-1) <= 1).mean() >= 0.999 and np.abs(rgb - g["rgb64"]).max() < 2e-3


def test_scene_uploaded_from_a_flat_image(gpu_ctx, scenes):
    """A scene that reaches the device as a flat image (built in another process, rt_scene_export_flat ->
    rt_scene_upload_flat, or rt_scene_import_flat -> rt_scene_upload) traces the reference's hits: the golden frame,
    the golden batch rays, and a mixed scene with spheres, a plane and a texture against the oracle."""
    img = scenes["teapot_normal"].export_flat()
    cam = rt.look_at(CAM[:3], CAM[3:])
    h = golden("hits_teapot_normal_96x54.npz")
    g = golden("rays_teapot_normal.npz")

    def check():
        ids, t, _, _ = gpu_ctx.trace_primary(cam, rt.default_params(width=96, height=54))
        assert eq(ids.ravel(), h["ids"].ravel()) and eq(np.where(ids >= 0, t, 1e9).ravel(), h["t"].ravel())
        ids, t = gpu_ctx.trace_closest(g["org_axis"], g["dir_axis"])
        assert eq(ids, g["ids_axis"]) and eq(t, g["t_axis"])
        ids32, t32 = gpu_ctx.trace_closest(g["org"], g["dir"], precision=rt.RT_F32)
        return ids32, t32

    gpu_ctx.upload(scenes["water"])                 # (something else is resident before each form)
    gpu_ctx.upload_flat(img)
    a = check()
    gpu_ctx.upload(scenes["water"])
    c = rt.Scene()
    c.import_flat(img)
    gpu_ctx.upload(c)
    b = check()
    gpu_ctx.upload(scenes["teapot_normal"])
    ref = check()
    assert eq(a[0], ref[0]) and eq(a[1], ref[1]) and eq(b[0], ref[0]) and eq(b[1], ref[1])

    # the full pixel path on an imported scene with every kind of record: same frame as from the scene itself
    def fill(api):
        m = api.add_texture_bmp(os.path.join(DATA, "tex.bmp"))
        api.load_obj(os.path.join(DATA, "teapot_normal.obj"), (0.5, 0.5, 0.5, 0.0), 1.0, (0, -0.4, 0), True, m)
        api.add_sphere((0.3, 0.5, 0.5), 0.25, (1, 1, 1, 1.8), 1.5)
        api.add_sphere((0.3, 0.5, -0.5), 0.2, (1, 0.5, 0.5, 0.6), 1.0)
        api.add_plane((0.0, 1.0, 0.0), -0.9, (0.5, 0.5, 0.5, 0))
        api.add_cornell()
        api.build_bvh()
        return api
    s = fill(rt.Scene())
    s.emit_photons(4000, seed=7)
    gpu_ctx.photons_upload(s.photons_get(0))         # (the reference has no direct light: a frame without a photon map is black)
    p = rt.default_params(width=96, height=64, photon_num=4000, photon_use=50)
    gpu_ctx.upload(s)
    rgb0, bgr0, _ = gpu_ctx.render(cam, p)
    gpu_ctx.upload_flat(s.export_flat())
    rgb1, bgr1, _ = gpu_ctx.render(cam, p)
    # (two renders of one scene agree to the order of the per-pixel atomic additions, not bit for bit)
    assert np.abs(rgb0 - rgb1).max() <= 1e-12 and eq(bgr0, bgr1) and rgb0.max() > 0
    with pytest.raises(rt.RtError) as e:
        gpu_ctx.upload_flat(np.zeros(4096, np.uint8))
    assert e.value.code == -1
    rgb2, _, _ = gpu_ctx.render(cam, p)              # a refused image leaves the resident scene as it was
    assert np.abs(rgb0 - rgb2).max() <= 1e-12


@pytest.mark.parametrize("tag", ["teapot1", "test", "teapot_normal", "water"])
def test_occlusion_frame_equals_the_reference(gpu_ctx, tag):
    """The benchmark's frame -- one word per pixel, bit 30 set where the any-hit query between light and hit point found
    an occluder -- against the reference's own BVH::Shoot on the same segments (tests/golden/shadow_frames.txt): any
    occluder found is an occluder, so the early exit must not change a single bit of the 1920x1080 frame."""
    import ctypes
    with open(os.path.join(os.path.dirname(DATA), "shadow_frames.txt")) as f:
        gold = {l.split()[0]: l.split()[1:] for l in f}
    gpu_ctx.upload(build_scene(rt.Scene(), tag + ".obj"))
    w, h = 1920, 1080
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
    d_ids = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d_ids), ctypes.c_size_t(w * h * 4)) == 0
    gpu_ctx.trace_primary_dev(rt.look_at(CAM[:3], CAM[3:]), rt.default_params(width=w, height=h, shadow_rays=2), (0, h), d_ids, None, None)
    folded = np.zeros(w * h, np.int32)
    assert hip.hipMemcpy(folded.ctypes.data_as(ctypes.c_void_p), d_ids, ctypes.c_size_t(w * h * 4), 2) == 0
    assert hip.hipFree(d_ids) == 0
    n_sh = int(((folded >= 0) & ((folded & (1 << 30)) != 0)).sum())
    assert [id_map_hash(folded), str(n_sh)] == gold[tag]


def test_blocks_with_more_than_64k_of_lds(gpu_ctx, scenes, monkeypatch):
    """Blocks of 8 waves over a staged node image take more than 64 KiB of dynamic LDS (8 stacks + the node array), which
    a launch is only given once the kernel has been allowed it: every entry point that sizes its blocks this way -- batch
    rays, camera rays, the pixel path and photon emission -- must still launch and return the golden results."""
    monkeypatch.setenv("RT_BLOCK_WAVES", "12")
    monkeypatch.setenv("RT_TRAV_MODE", "plain")
    monkeypatch.setenv("RT_TRACE_GEOM", "1")
    gpu_ctx.upload(scenes["teapot_normal"])
    g = golden("rays_teapot_normal.npz")
    ids, t = gpu_ctx.trace_closest(g["org"], g["dir"])
    assert eq(ids, g["ids"]) and eq(t, g["t"])
    ids32, _ = gpu_ctx.trace_closest(g["org"], g["dir"], precision=rt.RT_F32)
    h = golden("hits_teapot_normal_96x54.npz")
    cam = rt.look_at(CAM[:3], CAM[3:])
    ids, t, _, _ = gpu_ctx.trace_primary(cam, rt.default_params(width=96, height=54))
    assert eq(ids.ravel(), h["ids"].ravel()) and eq(np.where(ids >= 0, t, 1e9).ravel(), h["t"].ravel())
    s = build_scene(rt.Scene(), "teapot_normal.obj", (1.0, 1.0, 1.0, 1.8), 1.5, (0, -0.7, 0))
    gpu_ctx.upload(s)
    p = rt.default_params(width=96, height=64, use_photon_map=0)
    big, _, st_big = gpu_ctx.render(cam, p)
    st_emit = s.emit_photons_gpu(gpu_ctx, 20000, seed=3)
    n_big = s.photon_count(0)
    s.photons_clear()
    for k in ("RT_BLOCK_WAVES", "RT_TRAV_MODE", "RT_TRACE_GEOM"):
        monkeypatch.delenv(k)
    gpu_ctx.upload(scenes["teapot_normal"])
    ref32, _ = gpu_ctx.trace_closest(g["org"], g["dir"], precision=rt.RT_F32)
    assert eq(ids32, ref32)
    gpu_ctx.upload(s)
    plain, _, st_plain = gpu_ctx.render(cam, p)
    assert eq(big, plain) and st_big.secondary_rays == st_plain.secondary_rays > 0
    s.emit_photons_gpu(gpu_ctx, 20000, seed=3)
    assert s.photon_count(0) == n_big > 20000 and st_emit.secondary_rays > 20000


def test_tile_order_is_chosen_by_measurement_and_changes_no_frame(gpu_ctx, scenes, monkeypatch):
    """The camera-ray kernel takes a frame's tiles in one of two orders; the library times both on the first launches of a view
    and keeps the faster.  Whatever it does, the id map is the reference's: with either order fixed, during the timed
    launches, and after it has settled; and once settled it stays settled for the view."""
    gpu_ctx.upload(scenes["teapot_normal"])
    cam = rt.look_at(CAM[:3], CAM[3:])
    g = golden_hashes()["teapot_normal"]
    p = rt.default_params(width=1920, height=1080, shadow_rays=2)
    plain = rt.default_params(width=1920, height=1080)
    for fixed in ("1", "2", "4"):
        monkeypatch.setenv("RT_TILE_STRATA", fixed)
        ids, _, _, st = gpu_ctx.trace_primary(cam, plain)
        assert st.tile_order == int(fixed) and id_map_hash(ids) == g["hash"]
        folded = gpu_ctx.trace_primary(cam, p)[0]
        assert id_map_hash(np.where(folded >= 0, folded & ~(1 << 30), folded)) == g["hash"]
        if fixed == "1":
            reference = folded
        assert eq(folded, reference)
    monkeypatch.delenv("RT_TILE_STRATA")
    seen = []
    for i in range(13):      # (every call here waits for its kernel: the view's first launch, then the eight timed ones)
        folded, _, _, st = gpu_ctx.trace_primary(cam, p)
        assert eq(folded, reference)
        seen.append(int(st.tile_order))
    assert seen[:9] == [1] + [1, 2] * 4 and seen[9] in (1, 2) and seen[9:] == [seen[9]] * 4
    print("tile order settled on", seen[9], "strata for the teapot frame with occlusion queries")
    # another view is measured anew, from its second launch on; the first one keeps its order; a camera that moves stays plain
    cam2 = rt.look_at((2.4, -0.8, -1.0), (0, -0.5, 0))
    assert [int(gpu_ctx.trace_primary(cam2, p)[3].tile_order) for _ in range(4)] == [1, 1, 2, 1]
    assert int(gpu_ctx.trace_primary(cam, p)[3].tile_order) == seen[9]
    for i in range(1, 7):
        moving = rt.look_at((2.4 + 0.01 * i, -0.8, -1.0), (0, -0.5, 0))
        assert int(gpu_ctx.trace_primary(moving, p)[3].tile_order) == 1
    # single precision has a third order to measure (two tiles per wave): the id map is the plain order's in all three
    pf = rt.default_params(width=1920, height=1080, shadow_rays=2, precision=rt.RT_F32)
    monkeypatch.setenv("RT_TILE_ORDER", "1")
    ref32 = gpu_ctx.trace_primary(cam, pf)[0]
    for fixed in ("2", "3"):
        monkeypatch.setenv("RT_TILE_ORDER", fixed)
        ids32, _, _, st = gpu_ctx.trace_primary(cam, pf)
        assert st.tile_order == int(fixed) and eq(ids32, ref32)
    monkeypatch.delenv("RT_TILE_ORDER")
    seen32 = [int(gpu_ctx.trace_primary(cam, pf)[3].tile_order) for _ in range(15)]
    assert seen32[:13] == [1] + [1, 2, 3] * 4 and seen32[13] in (1, 2, 3) and seen32[14] == seen32[13]
    # (an odd frame: the two-tile form pads its grid)
    small = rt.default_params(width=200, height=136, precision=rt.RT_F32)
    monkeypatch.setenv("RT_TILE_ORDER", "1")
    ref_small = gpu_ctx.trace_primary(cam, small)[0]
    monkeypatch.setenv("RT_TILE_ORDER", "3")
    assert eq(gpu_ctx.trace_primary(cam, small)[0], ref_small)
