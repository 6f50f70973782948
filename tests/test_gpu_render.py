"""GPU parity of the shading path: photon k-NN gather + radiance estimate, the bounce queue
(reflection / refraction), textures, smooth normals, anti-aliasing, row tiling and the quantiser,
through the C ABI, against the CPU oracle and the golden vectors of the compiled reference.

Tolerances (stated by the north star): shaded 8-bit pixels within 1 LSB per channel on >= 99.9 % of
pixels; photon-mapped pixels additionally within an RMSE of 1.0 8-bit level (measured: < 0.1)."""
import os

import numpy as np
import pytest

import raytracer_amd as rt
from conftest import CAM, DATA, build_scene, golden
from oracle.bindings import Oracle

pytestmark = pytest.mark.gpu


def u8(rgb):
    return np.floor(np.clip(rgb, 0, 1) * 255).astype(np.int32)


def sort_rows(a):
    return a[np.lexsort(a.T[::-1])]


def assert_image_close(got_rgb, ref_rgb, frac=0.999, rmse_max=1.0, what=""):
    a, b = u8(got_rgb), u8(ref_rgb)
    diff = np.abs(a - b)
    within = (diff.max(axis=-1) <= 1).mean()
    rmse = float(np.sqrt(((a - b) ** 2).mean()))
    assert within >= frac, f"{what}: only {within:.5f} of pixels within 1 LSB (rmse {rmse:.4f})"
    assert rmse <= rmse_max, f"{what}: rmse {rmse:.4f} 8-bit levels"
    return within, rmse


@pytest.fixture(scope="module")
def cornell_photons(gpu_ctx):
    g = golden("photons_cornell.npz")
    return g


def test_knn_counts_and_radius(gpu_ctx, cornell_photons):
    g = cornell_photons
    gpu_ctx.photons_upload(g["photons"])
    for k in (8, 50, 1000):
        cnt, r2 = gpu_ctx.knn_radius(k, g["centers"])
        gold_cnt = g[f"cnt{k}"]
        d2 = g[f"d2_{k}"]
        gold_r2 = np.where(gold_cnt > 0, d2[np.arange(len(gold_cnt)), np.maximum(gold_cnt - 1, 0)], -1.0)
        # float32 distances: a photon within 1e-7 of the 0.28 radius may fall on the other side
        assert (np.abs(cnt - gold_cnt) <= 1).all() and (cnt == gold_cnt).mean() > 0.99, k
        same = cnt == gold_cnt
        assert np.allclose(r2[same], gold_r2[same], rtol=2e-5, atol=1e-9), k


def test_knn_equals_full_sort_on_gpu(gpu_ctx, cornell_photons):
    """KDTreeSelfTest's property (Kdtree.cpp:161-208): the k-th distance equals that of a full sort."""
    g = cornell_photons
    gpu_ctx.photons_upload(g["photons"])
    pos = g["photons"][:, :3].astype(np.float32).astype(np.float64)
    rng = np.random.default_rng(2)
    centers = rng.uniform(-1, 1, (64, 3)).astype(np.float32).astype(np.float64)
    for k in (1, 17, 300, 1000):
        cnt, r2 = gpu_ctx.knn_radius(k, centers)
        for q in range(len(centers)):
            dd = np.sort(((pos - centers[q]) ** 2).sum(1))
            within = dd[dd < np.float32(0.28 * 0.28)][:k]
            assert abs(cnt[q] - len(within)) <= 1
            if cnt[q] == len(within) and len(within):
                assert abs(r2[q] - within[-1]) <= 1e-5 * within[-1] + 1e-9


def test_radiance_estimate_matches_reference(gpu_ctx, cornell_photons):
    """getColorAt on diffuse hits (main.cpp:260-295) against the compiled reference's output."""
    g = cornell_photons
    gpu_ctx.photons_upload(g["photons"])
    orc = build_scene(Oracle(), None)
    hit_pos = g["shade_pos"] + g["shade_dir"] * 1e-6
    nrm = orc.obj_normal(g["shade_obj"], hit_pos)
    col = orc.obj_color(g["shade_obj"], hit_pos)[:, :3]
    p = rt.default_params()
    rgb, cnt = gpu_ctx.knn_estimate(p, hit_pos, nrm, col)
    ref = g["shade_rgb"]
    assert np.abs(rgb - ref).max() < 2e-3
    assert_image_close(rgb, ref, what="radiance estimate")
    rel = np.abs(rgb - ref)[ref > 0.05] / ref[ref > 0.05]
    assert np.median(rel) < 1e-5


@pytest.mark.parametrize("tag", ["diffuse", "mirror", "glass"])
def test_render_golden_head_semantics(gpu_ctx, tag):
    """64x64 frames the compiled reference shaded with its own getColorAt (HEAD recursion)."""
    g = golden(f"render_{tag}_64.npz")
    s = build_scene(rt.Scene(), "teapot_normal.obj", tuple(g["rgbs"]), float(g["ior"]), (0, -0.7, 0))
    n = s.emit_photons(3000, seed=12345)
    # host emission reproduces the reference's photon list (the golden is in the kd-tree's sorted order)
    assert np.array_equal(sort_rows(s.photons_get()), sort_rows(g["photons"]))
    gpu_ctx.upload(s)
    gpu_ctx.photons_upload(s.photons_get())
    p = rt.default_params(width=64, height=64, head_quirk=1)
    rgb, bgr, st = gpu_ctx.render(rt.look_at(CAM[:3], CAM[3:]), p)
    within, rmse = assert_image_close(rgb, g["rgb"], what=tag)
    assert np.array_equal(bgr[..., ::-1].astype(np.int32), u8(rgb))   # quantiser = floor(v * 255), BGR order
    assert st.primary_rays == 64 * 64


@pytest.mark.parametrize("tag,mesh", [("mirror", "teapot_normal.obj"), ("glass", "teapot_normal.obj"), ("water", "water.obj")])
def test_render_golden_intended_recursion(gpu_ctx, tag, mesh):
    """The default mode (head_quirk = 0: a bounce is continued on the object it hit) against 64x64 frames that the
    reference's own Shoot / getColor / getNormalAt / computeReflectionDir / getColorAt produced when driven with that
    recursion (HEAD's getColorAt passes the old object on, main.cpp:198,249, and shades mirrors and glass black)."""
    g = golden(f"render_{tag}_intended_64.npz")
    s = build_scene(rt.Scene(), mesh, tuple(g["rgbs"]), float(g["ior"]), tuple(g["off"]))
    s.emit_photons(3000, seed=12345)
    assert np.array_equal(sort_rows(s.photons_get()), sort_rows(g["photons"]))
    gpu_ctx.upload(s)
    gpu_ctx.photons_upload(s.photons_get())
    p = rt.default_params(width=64, height=64)
    assert p.head_quirk == 0
    rgb, bgr, st = gpu_ctx.render(rt.look_at(CAM[:3], CAM[3:]), p)
    within, rmse = assert_image_close(rgb, g["rgb"], what=tag + " (recursion on the new hit)")
    assert np.abs(rgb - g["rgb"]).max() < 2e-3
    assert st.secondary_rays > 0
    mesh_px = gpu_ctx.trace_primary(rt.look_at(CAM[:3], CAM[3:]), rt.default_params(width=64, height=64))[0] < s.size() - 10
    assert g["rgb"].reshape(-1, 3)[mesh_px].max() > 0.02                      # the reflecting / refracting mesh is not black


def test_render_golden_textures_spheres_aa(gpu_ctx):
    """Textured smooth-normal triangles + the reference's mirror and glass spheres at aadepth 1, 2 and 4 (main.cpp:771-787:
    aadepth^2 samples of which aadepth are distinct) against frames the reference's own functions produced."""
    from test_oracle_golden import tex_spheres_scene
    g = golden("render_texspheres_intended.npz")
    s = rt.Scene()
    tex_spheres_scene(s, s.add_texture_bmp(DATA + "/tex.bmp"))
    s.emit_photons(3000, seed=3)
    assert np.array_equal(sort_rows(s.photons_get()), sort_rows(g["photons"]))
    gpu_ctx.upload(s); gpu_ctx.photons_upload(s.photons_get())
    cam = rt.look_at(CAM[:3], CAM[3:])
    for (w, h, aa) in [(96, 72, 1), (80, 60, 2), (80, 60, 4)]:
        rgb, bgr, st = gpu_ctx.render(cam, rt.default_params(width=w, height=h, aadepth=aa))
        assert_image_close(rgb, g[f"rgb_{w}x{h}_aa{aa}"], what=f"textured + spheres {w}x{h} aadepth {aa}")
        assert st.primary_rays == w * h * aa


@pytest.mark.parametrize("tag,rgbs,ior", [("mirror", (1.0, 1.0, 1.0, 0.9), 1.5), ("glass", (1.0, 1.0, 1.0, 1.8), 1.5),
                                          ("water", (0.26, 0.84, 1.0, 1.8), 1.33)])
def test_render_intended_recursion(gpu_ctx, tag, rgbs, ior):
    """Recursion on the newly hit object (head_quirk = 0) against the oracle's same mode."""
    mesh, off = ("water.obj", (0, 0, 0)) if tag == "water" else ("teapot_normal.obj", (0, -0.7, 0))
    s = build_scene(rt.Scene(), mesh, rgbs, ior, off)
    orc = build_scene(Oracle(), mesh, rgbs, ior, off)
    s.emit_photons(4000, seed=99)
    ph = s.photons_get()
    orc.photons_set(np.c_[ph, np.zeros(len(ph))])
    orc.build_kdtree()
    gpu_ctx.upload(s)
    gpu_ctx.photons_upload(ph)
    w, h = 96, 72
    p = rt.default_params(width=w, height=h)
    rgb, bgr, st = gpu_ctx.render(rt.look_at(CAM[:3], CAM[3:]), p)
    orc.params.head_quirk = 0
    ref, nrays = orc.render(w, h, CAM, threads=8)
    assert_image_close(rgb, ref, what=tag)
    # ray accounting: the oracle counts every Shoot call, including those spawned at the last level
    # whose result is black by construction; the device skips those
    assert st.primary_rays == w * h and 0 < st.secondary_rays <= nrays - w * h
    assert st.gather_queries > 0


def test_render_textures_spheres_and_aa(gpu_ctx):
    """Textured smooth-normal triangles + mirror / glass spheres, aadepth 2 (reference sample pattern)."""
    def fill(api, tex):
        r2 = np.random.default_rng(5)
        for i in range(30):
            c = r2.uniform(-0.7, 0.7, 3)
            A, C, B = c + r2.uniform(-0.35, 0.35, (3, 3))
            o = api.add_triangle(A, C, B, (0.8, 0.7, 0.6, 0.0), 1.3)
            nn = r2.normal(size=(3, 3)); nn /= np.linalg.norm(nn, axis=1)[:, None]
            tt = r2.uniform(-1, 2, (3, 2))
            if hasattr(api, "last_set_normals"):
                api.last_set_normals(nn[0], nn[1], nn[2]); api.last_set_texture(tt[0], tt[1], tt[2], True, tex)
            else:
                api.set_normals(o, nn[0], nn[1], nn[2]); api.set_texture(o, tt[0], tt[1], tt[2], True, tex)
        api.add_sphere((0.3, -0.4, -0.4), 0.3, (1.0, 1.0, 1.0, 0.9), 1.5)     # the reference's two spheres (main.cpp:623-668)
        api.add_sphere((0.3, -0.7, 0.4), 0.3, (1.0, 1.0, 1.0, 1.8), 1.5)
        api.add_cornell()
        api.build_bvh()
    s = rt.Scene(); fill(s, s.add_texture_bmp(DATA + "/tex.bmp"))
    orc = Oracle(); fill(orc, orc.texture_read(DATA + "/tex.bmp"))
    s.emit_photons(3000, seed=3)
    ph = s.photons_get()
    orc.photons_set(np.c_[ph, np.zeros(len(ph))]); orc.build_kdtree()
    gpu_ctx.upload(s); gpu_ctx.photons_upload(ph)
    cam = rt.look_at(CAM[:3], CAM[3:])
    for aa in (1, 2, 3, 4, 8):
        w, h = 80, 60
        p = rt.default_params(width=w, height=h, aadepth=aa)
        rgb, bgr, st = gpu_ctx.render(cam, p)
        ref, _ = orc.render(w, h, CAM, aadepth=aa, threads=8)
        assert_image_close(rgb, ref, what=f"aadepth {aa}")
        assert st.primary_rays == w * h * aa


@pytest.mark.parametrize("aadepth", [1, 2, 4, 8])
def test_row_bands_and_interleave_compose_the_frame(gpu_ctx, aadepth):
    """Blocks of rows, and the bands of rows a rank of several renders, compose the frame -- also with the sample
    columns of anti-aliasing, which the waves of a tile share among themselves in another way per `aadepth`."""
    s = build_scene(rt.Scene(), "teapot_normal.obj", (1.0, 1.0, 1.0, 1.8), 1.5, (0, -0.7, 0))
    s.emit_photons(2000, seed=1)
    gpu_ctx.upload(s); gpu_ctx.photons_upload(s.photons_get())
    cam = rt.look_at(CAM[:3], CAM[3:])
    w, h = 64, 48
    kw = dict(width=w, height=h, aadepth=aadepth)
    full, full8, _ = gpu_ctx.render(cam, rt.default_params(**kw))
    top, _, _ = gpu_ctx.render(cam, rt.default_params(**kw), rows=(0, 20))
    bot, _, _ = gpu_ctx.render(cam, rt.default_params(**kw), rows=(20, 48))
    # (another split of the rows groups the gather requests differently; a group adds four photons' terms in float
    # before they join its double sum, so the frames agree to float rounding of a term, not to the last bit)
    assert np.abs(np.concatenate([top, bot]) - full).max() < 5e-7
    parts = [gpu_ctx.render(cam, rt.default_params(band_rows=8, n_parts=3, part=r, **kw))[0] for r in range(3)]
    stacked = np.stack([p.reshape(h // 24, 8, w, 3) for p in parts], axis=1).reshape(h, w, 3)
    assert np.abs(stacked - full).max() < 5e-7


def test_render_f32_instantiation(gpu_ctx):
    """FP32 traversal + f64 shading on a glass teapot (every pixel of it runs through refraction
    chains): not the parity instantiation; at this small size >= 99 % of the pixels stay within
    1 LSB of the f64 frame (silhouette pixels may pick the neighbouring object)."""
    s = build_scene(rt.Scene(), "teapot_normal.obj", (1.0, 1.0, 1.0, 1.8), 1.5, (0, -0.7, 0))
    s.emit_photons(3000, seed=12345)
    gpu_ctx.upload(s); gpu_ctx.photons_upload(s.photons_get())
    cam = rt.look_at(CAM[:3], CAM[3:])
    a, _, _ = gpu_ctx.render(cam, rt.default_params(width=128, height=96))
    b, _, _ = gpu_ctx.render(cam, rt.default_params(width=128, height=96, precision=rt.RT_F32))
    assert_image_close(b, a, frac=0.99, rmse_max=3.0, what="f32 vs f64")


def _d2_f32(pos32, c32):
    d = pos32 - c32
    return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]


@pytest.mark.parametrize("front_cap", [None, "32"])
def test_knn_stress_paths(gpu_ctx, front_cap, monkeypatch):
    """Large radius over a volume-filling cloud (many subtrees inside the bound: the frontier
    overflows into the serial walk) plus thousands of coincident photons (crowded histogram bins:
    the select falls back to its radix passes, ties at the k-th distance).  Count and k-th distance
    must equal a full sort in the same float32 arithmetic.  The second run shrinks the frontier to 32
    entries (a launch-time setting of the library) so that the serial walk is certainly reached."""
    if front_cap:
        monkeypatch.setenv("RT_GATHER_FRONT_CAP", front_cap)
    rng = np.random.default_rng(11)
    n = 200_000
    ph = np.zeros((n, 9))
    ph[:, :3] = rng.uniform(-1, 1, (n, 3))
    ph[:6000, :3] = np.array([0.25, -0.5, 0.125])            # coincident cluster
    ph[6000:9000, :3] = np.array([0.25, -0.5, 0.125]) + rng.normal(0, 1e-4, (3000, 3))
    ph[:, 3:6] = 1.0
    ph[:, 6:9] = np.array([0.0, -1.0, 0.0])
    gpu_ctx.photons_upload(ph)
    pos32 = ph[:, :3].astype(np.float32)
    centers = np.concatenate([rng.uniform(-1, 1, (24, 3)), np.array([[0.25, -0.5, 0.125], [0.26, -0.5, 0.125], [0.0, 0.0, 0.0]])])
    centers = centers.astype(np.float32).astype(np.float64)
    for k, radius in ((1000, 5.0), (1000, 0.28), (64, 5.0), (1792, 0.5)):
        cnt, r2 = gpu_ctx.knn_radius(k, centers, max_dist=radius)
        lim = np.float32(radius * radius)
        for q in range(len(centers)):
            dd = np.sort(_d2_f32(pos32, centers[q].astype(np.float32)))
            within = dd[dd < lim][:k]
            assert cnt[q] == len(within), (k, radius, q, cnt[q], len(within))
            if len(within):
                assert np.float32(r2[q]) == within[-1], (k, radius, q, r2[q], within[-1])


def test_render_subsurface(gpu_ctx):
    """Subsurface material (special < 0, main.cpp:297-388) against the compiled reference's output:
    surface estimate + volume gather + two-pass march.  The march's step lengths come from a
    counter-based generator here and from the shared mt19937 in the reference, so the comparison is
    statistical (RMSE), as for every photon-mapped pixel."""
    g = golden("render_sss_64.npz")
    s = build_scene(rt.Scene(), "teapot_normal.obj", tuple(g["rgbs"]), float(g["ior"]), (0, -0.7, 0))
    s.emit_photons(20000, seed=12345)
    assert np.array_equal(sort_rows(s.photons_get(0)), sort_rows(g["photons"]))
    assert np.array_equal(sort_rows(s.photons_get(1)), sort_rows(g["volume_photons"]))
    gpu_ctx.upload(s)
    gpu_ctx.photons_upload(s.photons_get(0))
    gpu_ctx.photons_upload(s.photons_get(1), which=1)
    try:
        p = rt.default_params(width=64, height=64, head_quirk=1)
        rgb, bgr, st = gpu_ctx.render(rt.look_at(CAM[:3], CAM[3:]), p)
    finally:
        gpu_ctx.photons_upload(np.zeros((0, 9)), which=1)
    ref = g["rgb"]
    a, b = u8(rgb), u8(ref)
    within = (np.abs(a - b).max(-1) <= 1).mean()
    rmse = float(np.sqrt(((a - b) ** 2).mean()))
    print(f"subsurface: within 1 LSB {within:.4f}, rmse {rmse:.3f}, max {np.abs(a - b).max()}")
    assert within >= 0.99 and rmse <= 1.0


def test_subsurface_terms_in_groups_equal_per_request(gpu_ctx, monkeypatch):
    """The group gather answers the surface estimate of subsurface requests and their three volume terms (main.cpp:310,
    346: at the hit point and along the march) and finishes the request; what it cannot take it hands on with the surface
    estimate in the record.  With every gather answered per request the frame is the same to the rounding of the float
    partial sums -- also with the volume terms alone per request, and with the per-request kernel only after the group
    kernel instead of alongside it (the hand-over of completed records)."""
    g = golden("render_sss_64.npz")
    s = build_scene(rt.Scene(), "teapot_normal.obj", tuple(g["rgbs"]), float(g["ior"]), (0, -0.7, 0))
    s.emit_photons(20000, seed=12345)
    gpu_ctx.upload(s)
    gpu_ctx.photons_upload(s.photons_get(0))
    gpu_ctx.photons_upload(s.photons_get(1), which=1)
    try:
        cam = rt.look_at(CAM[:3], CAM[3:])
        for quirk in (0, 1):
            p = rt.default_params(width=160, height=120, head_quirk=quirk, photon_use=200)
            grouped, _, st_g = gpu_ctx.render(cam, p)
            monkeypatch.setenv("RT_GATHER_MODE", "query")
            single, _, st_q = gpu_ctx.render(cam, p)
            monkeypatch.delenv("RT_GATHER_MODE")
            assert st_g.gather_queries == st_q.gather_queries > 5000
            assert grouped.max() > 0.05
            err = np.abs(grouped - single).max()
            print(f"subsurface, grouped terms vs per request (head_quirk {quirk}): max abs difference {err:.3e}, {st_g.gather_single} of {st_g.gather_queries} requests one per wave")
            assert err < 2e-6 * max(1.0, np.abs(single).max())
            for switch in ("RT_SSS_VOLUME_PER_REQUEST", "RT_GATHER_NO_ALONGSIDE"):
                monkeypatch.setenv(switch, "1")
                other, _, st_o = gpu_ctx.render(cam, p)
                monkeypatch.delenv(switch)
                err = np.abs(other - single).max()
                print(f"  {switch}: max abs difference {err:.3e}, {st_o.gather_single} one per wave")
                assert err < 2e-6 * max(1.0, np.abs(single).max())
                if switch == "RT_SSS_VOLUME_PER_REQUEST":      # (every subsurface request ends per request then; the groups finish many of them)
                    assert st_o.gather_single > 2 * st_g.gather_single
    finally:
        gpu_ctx.photons_upload(np.zeros((0, 9)), which=1)


def test_subsurface_full_frame_volume_terms_in_groups(gpu_ctx, monkeypatch):
    """The subsurface teapot at 1920x1080 with maps of 200 000 emitted photons and k = 1000: the frame with the volume terms
    answered in groups against the one with them answered per request (surface estimates in groups in both), every pixel."""
    s = rt.Scene()
    s.load_obj(os.path.join(DATA, "teapot_normal.obj"), (0.9, 0.7, 0.6, -1.0), 1.3, (0, -0.7, 0))
    s.add_cornell()
    s.build_bvh()
    gpu_ctx.upload(s)
    (ns, nv), _ = gpu_ctx.emit_photon_maps(200000, seed=12345)
    assert ns > 100000 and nv > 20000
    try:
        cam = rt.look_at((2.8, 0, 0), (0, 0, 0))
        p = rt.default_params(width=1920, height=1080, photon_num=200000, photon_use=1000)
        grouped, _, st_g = gpu_ctx.render(cam, p)
        monkeypatch.setenv("RT_SSS_VOLUME_PER_REQUEST", "1")
        single, _, st_q = gpu_ctx.render(cam, p)
        monkeypatch.delenv("RT_SSS_VOLUME_PER_REQUEST")
        err = np.abs(grouped - single).max()
        print(f"subsurface 1920x1080 k=1000: volume terms in groups vs per request, max abs difference {err:.3e}; requests one per wave {st_g.gather_single} vs {st_q.gather_single}; "
              f"device {st_g.kernel_ms:.2f} vs {st_q.kernel_ms:.2f} ms")
        assert st_q.gather_single > 50000 and st_g.gather_single < st_q.gather_single // 4
        assert err < 2e-6 * max(1.0, np.abs(single).max())
    finally:
        gpu_ctx.photons_upload(np.zeros((0, 9)), which=0)
        gpu_ctx.photons_upload(np.zeros((0, 9)), which=1)


def test_subsurface_without_surface_photons_in_reach(gpu_ctx, monkeypatch):
    """Subsurface hit points with no photon of the surface map within the search radius (the map holds the photons of one
    wall only): their surface estimate is zero, and their volume terms are still owed -- in both group forms as per request."""
    g = golden("render_sss_64.npz")
    s = build_scene(rt.Scene(), "teapot_normal.obj", tuple(g["rgbs"]), float(g["ior"]), (0, -0.7, 0))
    s.emit_photons(20000, seed=12345)
    gpu_ctx.upload(s)
    surface = s.photons_get(0)
    far = surface[surface[:, 0] < -0.95]
    assert 200 < len(far) < len(surface)
    gpu_ctx.photons_upload(far)
    gpu_ctx.photons_upload(s.photons_get(1), which=1)
    try:
        cam = rt.look_at(CAM[:3], CAM[3:])
        p = rt.default_params(width=160, height=120, photon_use=200)
        monkeypatch.setenv("RT_GATHER_MODE", "query")
        single, _, st_q = gpu_ctx.render(cam, p)
        monkeypatch.delenv("RT_GATHER_MODE")
        teapot = single[40:80, 60:100]
        assert teapot.max() > 0.01                 # (volume terms alone light the teapot)
        for form in ("stream", "tile"):
            monkeypatch.setenv("RT_GATHER_FORM", form)
            grouped, _, st_g = gpu_ctx.render(cam, p)
            monkeypatch.delenv("RT_GATHER_FORM")
            err = np.abs(grouped - single).max()
            print(f"subsurface without surface photons in reach, form {form}: max abs difference {err:.3e}, {st_g.gather_single} of {st_g.gather_queries} one per wave")
            assert err < 2e-6 * max(1.0, np.abs(single).max())
    finally:
        gpu_ctx.photons_upload(np.zeros((0, 9)), which=1)


def _glossy_scene(api):
    api.load_obj(os.path.join(DATA, "teapot_normal.obj"), (0.5, 0.5, 0.5, 0.0), 1.0, (0, -0.7, 0))
    api.add_sphere((0.0, 0.35, 0.55), 0.3, (1.0, 1.0, 1.0, 0.987654), 1.0)      # glossy mirror (main.cpp:640)
    api.add_sphere((0.0, 0.35, -0.55), 0.3, (1.0, 1.0, 1.0, 1.887654), 1.5)     # glossy glass (main.cpp:641)
    api.add_cornell()
    api.build_bvh()
    return api


def test_render_glossy_materials(gpu_ctx):
    """The 16-sample glossy loops of getColorAt (main.cpp:139-156, 177-257) against the oracle drawing from
    the same counter-based streams: quirks included (positions along the un-jittered direction, the in-loop
    normal flip, the in-loop scaling of the refraction sum)."""
    s = _glossy_scene(rt.Scene())
    orc = _glossy_scene(Oracle())
    gpu_ctx.upload(s)
    s.emit_photons(20000, seed=3)
    ph = s.photons_get(0)
    gpu_ctx.photons_upload(ph)
    orc.photons_set(np.c_[ph, np.zeros(len(ph))], 0)
    orc.build_kdtree()
    w, h = 160, 120
    kw = dict(width=w, height=h, photon_num=20000, photon_use=60)
    p = rt.default_params(rng_seed=777, **kw)
    rgb, bgr, st = gpu_ctx.render(rt.look_at(CAM[:3], CAM[3:]), p)
    orc.params.photon_num = 20000
    orc.params.photon_use = 60
    orc.params.counter_rng = 1
    orc.params.rng_seed = 777
    ref, nrays = orc.render(w, h, CAM, threads=8)
    # (the oracle also shoots the rays of the last level, whose hits shade black; the GPU does not trace those)
    assert st.primary_rays == w * h and 100000 < st.secondary_rays <= nrays - w * h
    a, b = u8(rgb), u8(ref)
    within = (np.abs(a - b).max(-1) <= 1).mean()
    print(f"glossy: within 1 LSB {within:.5f}, max {np.abs(a - b).max()}, rays {nrays}")
    assert within >= 0.999
    ids, _, _, _ = gpu_ctx.trace_primary(rt.look_at(CAM[:3], CAM[3:]), rt.default_params(width=w, height=h))
    balls = ((ids == 992) | (ids == 993)).reshape(h, w)
    assert balls.sum() > 500 and rgb[balls].max() > 0.05
    # another seed, another frame
    rgb2, _, _ = gpu_ctx.render(rt.look_at(CAM[:3], CAM[3:]), rt.default_params(rng_seed=778, **kw))
    assert np.abs(rgb2 - rgb)[balls].max() > 1e-3 and np.abs(rgb2 - rgb)[~balls].max() < 1e-9


def _glossy_panes_scene(api):
    api.load_obj(os.path.join(DATA, "teapot_normal.obj"), (0.5, 0.5, 0.5, 0.0), 1.0, (0, -0.7, 0))
    # two panes of glossy glass between the camera and the box: every pixel sees the first, and the rays that pass it
    # meet the second, whose reflections meet the first again
    api.add_plane((1.0, 0.0, 0.0), 2.0, (1.0, 1.0, 1.0, 1.887654))
    api.add_plane((1.0, 0.0, 0.0), 1.8, (1.0, 1.0, 1.0, 1.887654))
    api.add_cornell()
    api.build_bvh()
    return api


def test_render_glossy_glass_filling_the_frame(gpu_ctx, monkeypatch):
    """Glossy glass in every pixel at the default depth of 4: each hit spawns 16 + 16 rays (main.cpp:177-257) and the
    reflected ones meet the glass again, so a pixel ends in thousands of rays -- far more than the ray queues of a
    block of rows hold.  The frame is then rendered in parts (rows, then columns) instead of failing, and comes out
    as the oracle's and the same whatever the queues' size."""
    s = _glossy_panes_scene(rt.Scene())
    orc = _glossy_panes_scene(Oracle())
    gpu_ctx.upload(s)
    s.emit_photons(5000, seed=3)
    ph = s.photons_get(0)
    gpu_ctx.photons_upload(ph)
    orc.photons_set(np.c_[ph, np.zeros(len(ph))], 0)
    orc.build_kdtree()
    w, h = 48, 32
    p = rt.default_params(width=w, height=h, photon_num=5000, photon_use=20, rng_seed=99)
    assert p.tracing_depth == 4
    cam = rt.look_at(CAM[:3], CAM[3:])
    rgb, bgr, st = gpu_ctx.render(cam, p)
    ids, _, _, _ = gpu_ctx.trace_primary(cam, rt.default_params(width=w, height=h))
    assert len(np.unique(ids)) == 1 and ids.min() >= 0                              # the first pane, everywhere
    assert st.secondary_rays > 2000 * w * h                                          # far beyond 1024 slots per camera ray
    orc.params.photon_num = 5000
    orc.params.photon_use = 20
    orc.params.counter_rng = 1
    orc.params.rng_seed = 99
    ref, nrays = orc.render(w, h, CAM, threads=8)
    a, b = u8(rgb), u8(ref)
    within = (np.abs(a - b).max(-1) <= 1).mean()
    print(f"glossy panes: within 1 LSB {within:.5f}, max {np.abs(a - b).max()}, rays {st.secondary_rays} (oracle {nrays}), mean {b.mean():.2f}")
    assert b.max() > 10 and within >= 0.999
    # smaller queues: more parts, the same frame (sums of the same terms in another order)
    monkeypatch.setenv("RT_SHARD_CAP", "40000")
    rgb2, _, st2 = gpu_ctx.render(cam, p)
    monkeypatch.delenv("RT_SHARD_CAP")
    assert st2.secondary_rays == st.secondary_rays and st2.gather_queries == st.gather_queries
    assert np.abs(rgb2 - rgb).max() < 1e-9 * max(1.0, np.abs(rgb).max())
    # a frame whose sides are no multiples of the tile: the parts' last columns and rows are ragged
    pr = rt.default_params(width=50, height=37, photon_num=5000, photon_use=20, rng_seed=99)
    whole, _, st3 = gpu_ctx.render(cam, pr)
    monkeypatch.setenv("RT_SHARD_CAP", "40000")
    parts, _, st4 = gpu_ctx.render(cam, pr)
    monkeypatch.delenv("RT_SHARD_CAP")
    assert st4.secondary_rays == st3.secondary_rays > 2000 * 50 * 37 and np.abs(parts - whole).max() < 1e-9 * max(1.0, np.abs(whole).max())


@pytest.mark.parametrize("depth", [6, 8])
def test_render_deeper_than_the_default(gpu_ctx, monkeypatch, depth):
    """TRACING_DEPTH above the reference's 4: more levels than the queues are sized for in the worst case, so they are
    a budget and a block of rows that outgrows it is split and rendered again.  The frame is the oracle's at the same
    depth, and the same for small queues."""
    rgbs, ior, off = (1.0, 1.0, 1.0, 1.8), 1.5, (0, -0.7, 0)
    s = build_scene(rt.Scene(), "teapot_normal.obj", rgbs, ior, off)
    orc = build_scene(Oracle(), "teapot_normal.obj", rgbs, ior, off)
    gpu_ctx.upload(s)
    s.emit_photons(20000, seed=5)
    ph = s.photons_get(0)
    gpu_ctx.photons_upload(ph)
    orc.photons_set(np.c_[ph, np.zeros(len(ph))], 0)
    orc.build_kdtree()
    w, h = 96, 72
    p = rt.default_params(width=w, height=h, photon_num=20000, photon_use=50, tracing_depth=depth)
    cam = rt.look_at(CAM[:3], CAM[3:])
    rgb, _, st = gpu_ctx.render(cam, p)
    orc.params.photon_num = 20000
    orc.params.photon_use = 50
    orc.params.tracing_depth = depth
    ref, _ = orc.render(w, h, CAM, threads=8)
    assert_image_close(rgb, ref, what=f"glass teapot at depth {depth}")
    shallow, _, st4 = gpu_ctx.render(cam, rt.default_params(width=w, height=h, photon_num=20000, photon_use=50))
    assert st.secondary_rays > st4.secondary_rays and np.abs(rgb - shallow).max() > 1e-3       # the extra levels are there
    monkeypatch.setenv("RT_SHARD_CAP", "1024")
    small, _, st_small = gpu_ctx.render(cam, p)
    monkeypatch.delenv("RT_SHARD_CAP")
    assert st_small.secondary_rays == st.secondary_rays and np.abs(small - rgb).max() < 1e-9
    # (a part that overflowed is split before its photon gather is launched: only the parts that fit gather, once each)
    assert st_small.gather_queries == st.gather_queries


@pytest.mark.parametrize("head_quirk", [0, 1])
def test_no_photon_map_traces_the_same_rays(gpu_ctx, head_quirk):
    """Without a photon map the frame is black (the reference has no direct light) and hits that spawn no rays are not
    queued for shading; the rays traced are exactly those of the frame with a photon map, in both recursion modes."""
    s = _glossy_scene(rt.Scene())
    gpu_ctx.upload(s)
    s.emit_photons(5000, seed=3)
    gpu_ctx.photons_upload(s.photons_get(0))
    cam = rt.look_at(CAM[:3], CAM[3:])
    kw = dict(width=160, height=120, photon_num=5000, photon_use=20, rng_seed=5, head_quirk=head_quirk)
    lit, _, st1 = gpu_ctx.render(cam, rt.default_params(use_photon_map=1, **kw))
    dark, bgr, st0 = gpu_ctx.render(cam, rt.default_params(use_photon_map=0, **kw))
    assert st0.secondary_rays == st1.secondary_rays > 50000 and st0.primary_rays == st1.primary_rays
    assert st0.gather_queries == 0 and st1.gather_queries > 0
    assert (dark == 0).all() and (bgr == 0).all() and lit.max() > 0.05


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("which", ["teapot", "mixed", "water"])
def test_queue_tracer_scene_in_lds_same_frame(gpu_ctx, monkeypatch, precision, which):
    """Small scenes are traced from LDS by the queue tracer (nodes and primitive records staged once per block), scenes
    of a few thousand triangles with their nodes in LDS, larger ones through L1: the same walk on the same records, so
    switching the staging off changes nothing -- the same rays and the same frame up to the order of the per-pixel
    sums.  Triangles only and mixed primitives (the two instantiations of the leaf test), the water surface (nodes
    only), both precisions."""
    if which == "mixed":
        s = _random_scene(rt.Scene(), 21)
    else:
        s = rt.Scene()
        if which == "teapot":
            s.load_obj(os.path.join(DATA, "teapot_normal.obj"), (1.0, 1.0, 1.0, 1.8), 1.5, (0, -0.7, 0))
        else:
            s.load_obj(os.path.join(DATA, "water.obj"), (0.26, 0.84, 1.0, 1.8), 1.33, (0, 0, 0))
        s.add_cornell()
        s.build_bvh()
    gpu_ctx.upload(s)
    s.emit_photons(4000, seed=9)
    gpu_ctx.photons_upload(s.photons_get(0))
    cam = rt.look_at(CAM[:3], CAM[3:])
    p = rt.default_params(width=320, height=180, photon_num=4000, photon_use=30, rng_seed=4,
                          precision=rt.RT_F64 if precision == "f64" else rt.RT_F32)
    lds, _, st_lds = gpu_ctx.render(cam, p)
    monkeypatch.setenv("RT_NO_SCENE_LDS", "1")
    l1, _, st_l1 = gpu_ctx.render(cam, p)
    monkeypatch.delenv("RT_NO_SCENE_LDS")
    assert st_lds.secondary_rays == st_l1.secondary_rays > 10000 and st_lds.gather_queries == st_l1.gather_queries
    assert lds.max() > 0.05 and np.abs(lds - l1).max() < 1e-9


def _random_scene(api, seed):
    """Spheres, loose triangles and a plane of random materials (diffuse, mirror, glass, glossy mirror, glossy glass)
    inside the Cornell box."""
    r = np.random.default_rng(seed)
    specials = [0.0, 0.0, 0.6, 1.8, 0.987654, 1.887654]
    for _ in range(int(r.integers(3, 7))):
        c = r.uniform(-0.6, 0.6, 3)
        sp = specials[int(r.integers(0, len(specials)))]
        api.add_sphere(tuple(c), float(r.uniform(0.12, 0.3)), (float(r.uniform(0.3, 1)), float(r.uniform(0.3, 1)), float(r.uniform(0.3, 1)), sp),
                       float(r.uniform(1.2, 1.7)))
    for _ in range(int(r.integers(2, 6))):
        A = r.uniform(-0.8, 0.8, 3)
        B, C = A + r.uniform(-0.5, 0.5, 3), A + r.uniform(-0.5, 0.5, 3)
        sp = specials[int(r.integers(0, 4))]
        api.add_triangle(tuple(A), tuple(C), tuple(B), (float(r.uniform(0.3, 1)), float(r.uniform(0.3, 1)), float(r.uniform(0.3, 1)), sp), 1.5)
    api.add_plane((0.0, 1.0, 0.0), -0.95, (0.7, 0.7, 0.7, 0.0))
    api.add_cornell()
    api.build_bvh()
    return api


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_render_random_scenes_match_oracle(gpu_ctx, seed):
    """Random mixes of the primitive kinds and material classes, anti-aliased, at the default depth: the frame is the
    restatement's (same counter-based draws for the glossy loops)."""
    s = _random_scene(rt.Scene(), seed)
    orc = _random_scene(Oracle(), seed)
    gpu_ctx.upload(s)
    s.emit_photons(8000, seed=seed)
    ph = s.photons_get(0)
    gpu_ctx.photons_upload(ph)
    orc.photons_set(np.c_[ph, np.zeros(len(ph))], 0)
    orc.build_kdtree()
    w, h, aa = 72, 54, 2
    p = rt.default_params(width=w, height=h, aadepth=aa, photon_num=8000, photon_use=40, rng_seed=seed)
    rgb, bgr, st = gpu_ctx.render(rt.look_at(CAM[:3], CAM[3:]), p)
    orc.params.photon_num = 8000
    orc.params.photon_use = 40
    orc.params.counter_rng = 1
    orc.params.rng_seed = seed
    ref, nrays = orc.render(w, h, CAM, aadepth=aa, threads=8)
    a, b = u8(rgb), u8(ref)
    within = (np.abs(a - b).max(-1) <= 1).mean()
    print(f"random scene {seed}: within 1 LSB {within:.5f}, max {np.abs(a - b).max()}, rays {st.secondary_rays}, mean level {b.mean():.1f}")
    assert b.max() > 20 and within >= 0.999


@pytest.mark.parametrize("aadepth,dof", [(1, 8), (2, 3)])
def test_render_depth_of_field(gpu_ctx, aadepth, dof):
    """Depth of field (main.cpp:792-826, DOF_SAMPLE > 1): lens positions from the counter-based stream, the
    hit position taken along the un-jittered direction as the reference does."""
    rgbs, ior, off = (1.0, 1.0, 1.0, 1.8), 1.5, (0, -0.7, 0)
    s = build_scene(rt.Scene(), "teapot_normal.obj", rgbs, ior, off)
    orc = build_scene(Oracle(), "teapot_normal.obj", rgbs, ior, off)
    gpu_ctx.upload(s)
    s.emit_photons(20000, seed=3)
    ph = s.photons_get(0)
    gpu_ctx.photons_upload(ph)
    orc.photons_set(np.c_[ph, np.zeros(len(ph))], 0)
    orc.build_kdtree()
    w, h = 128, 96
    kw = dict(width=w, height=h, photon_num=20000, photon_use=60, aadepth=aadepth)
    p = rt.default_params(dof_samples=dof, focal_len=2.4, aperture=0.15, **kw)
    rgb, bgr, st = gpu_ctx.render(rt.look_at(CAM[:3], CAM[3:]), p)
    orc.params.photon_num = 20000
    orc.params.photon_use = 60
    orc.params.counter_rng = 1
    orc.params.dof_samples = dof
    orc.params.focal_len = 2.4
    orc.params.aperture = 0.15
    ref, nrays = orc.render(w, h, CAM, aadepth=aadepth, threads=8)
    assert st.primary_rays == w * h * aadepth * aadepth * dof
    assert 0 < st.secondary_rays <= nrays - st.primary_rays
    within, rmse = assert_image_close(rgb, ref, what=f"dof {dof} aa {aadepth}")
    sharp, _, _ = gpu_ctx.render(rt.look_at(CAM[:3], CAM[3:]), rt.default_params(**kw))
    assert rgb.max() > 0.1 and np.abs(sharp - rgb).mean() > 1e-4


def _emission_summary(ph):
    """Count, mean power per channel, per-axis position histogram (8 bins over the box)."""
    hist = np.stack([np.histogram(ph[:, ax], bins=8, range=(-1.0001, 1.0001))[0] for ax in range(3)]).astype(np.float64)
    mean_power = ph[:, 3:6].mean(0) if len(ph) else np.zeros(3)
    return len(ph), mean_power, hist / max(1, len(ph))


@pytest.mark.parametrize("tag,rgbs,ior", [("glass", (1.0, 1.0, 1.0, 1.8), 1.5), ("mirror", (1.0, 1.0, 1.0, 0.9), 1.5),
                                          ("subsurface", (0.9, 0.7, 0.6, -1.0), 1.3)])
def test_gpu_emission_statistics(gpu_ctx, tag, rgbs, ior, monkeypatch):
    """photonEmission on the GPU (per-photon random streams) against the serial host emission that
    reproduces the reference's list: same distribution -- photons stored per emitted photon, mean
    power, where the photons land -- within sampling noise; identical lists for identical arguments."""
    s = build_scene(rt.Scene(), "teapot_normal.obj", rgbs, ior, (0, -0.7, 0))
    gpu_ctx.upload(s)
    n = 60000
    s.emit_photons(n, seed=5)
    host = [s.photons_get(0), s.photons_get(1)]
    s.photons_clear()
    st = s.emit_photons_gpu(gpu_ctx, n, seed=5)
    dev = [s.photons_get(0), s.photons_get(1)]
    s.photons_clear()
    s.emit_photons_gpu(gpu_ctx, n, seed=5)
    assert np.array_equal(s.photons_get(0), dev[0]) and np.array_equal(s.photons_get(1), dev[1])   # launch-independent order
    # which photons share a wave (the launch orders them by starting direction) does not change the list
    s.photons_clear()
    monkeypatch.setenv("RT_EMIT_UNORDERED", "1")
    st_plain = s.emit_photons_gpu(gpu_ctx, n, seed=5)
    monkeypatch.delenv("RT_EMIT_UNORDERED")
    assert st_plain.secondary_rays == st.secondary_rays
    assert np.array_equal(s.photons_get(0), dev[0]) and np.array_equal(s.photons_get(1), dev[1])
    # nor does re-ordering the survivors between bounces (a comparison switch: measured slower, the sorts cost more than they return)
    s.photons_clear()
    monkeypatch.setenv("RT_EMIT_RESORT", "1")
    st_sorted = s.emit_photons_gpu(gpu_ctx, n, seed=5)
    monkeypatch.delenv("RT_EMIT_RESORT")
    assert st_sorted.secondary_rays == st.secondary_rays
    assert np.array_equal(s.photons_get(0), dev[0]) and np.array_equal(s.photons_get(1), dev[1])
    assert st.secondary_rays >= n
    for which in (0, 1):
        nh, ph_, hh = _emission_summary(host[which])
        nd, pd_, hd = _emission_summary(dev[which])
        if which == 1 and tag != "subsurface":
            assert nh == 0 and nd == 0
            continue
        assert nh > 1000
        assert abs(nd - nh) <= 5 * np.sqrt(nh) + 0.01 * nh, (tag, which, nh, nd)
        assert np.abs(pd_ - ph_).max() < 0.02, (tag, which, ph_, pd_)
        assert np.abs(hd - hh).max() < 0.02, (tag, which)
        # photons lie on surfaces the host's photons lie on: every GPU photon has a host photon nearby
    d = np.linalg.norm(dev[0][:2000, None, :3] - host[0][None, ::7, :3], axis=-1).min(1)
    assert np.median(d) < 0.05


def test_full_frame_glass_against_oracle(gpu_ctx):
    """Configuration C3 at its full 3840x2160: glass teapot with smooth normals, recursion depth 4, the
    whole frame against the CPU oracle (recursion on the newly hit object).  The renderer has no direct
    lighting -- without photons every pixel is black -- so a small photon map (k = 30, which the oracle
    gathers in seconds) gives the bounce paths something to carry."""
    import os
    rgbs, ior, off = (1.0, 1.0, 1.0, 1.8), 1.5, (0, -0.7, 0)
    s = build_scene(rt.Scene(), "teapot_normal.obj", rgbs, ior, off)
    orc = build_scene(Oracle(), "teapot_normal.obj", rgbs, ior, off)
    gpu_ctx.upload(s)
    s.emit_photons(10000, seed=5)
    ph = s.photons_get()
    gpu_ctx.photons_upload(ph)
    orc.photons_set(np.c_[ph, np.zeros(len(ph))])
    orc.build_kdtree()
    w, h = 3840, 2160
    p = rt.default_params(width=w, height=h, photon_num=10000, photon_use=30)
    rgb, bgr, st = gpu_ctx.render(rt.look_at(CAM[:3], CAM[3:]), p)
    orc.params.head_quirk = 0
    orc.params.photon_num = 10000
    orc.params.photon_use = 30
    ref, nrays = orc.render(w, h, CAM, threads=min(16, os.cpu_count() or 1))
    within, rmse = assert_image_close(rgb, ref, what="C3 3840x2160")
    assert ref.max() > 0.2 and (ref.sum(-1) > 0).mean() > 0.5      # a lit frame, not a black one
    assert st.primary_rays == w * h and st.secondary_rays > 100000


def test_large_frame_photon_map_against_oracle(gpu_ctx):
    """Configuration C4 scaled to what the CPU oracle finishes in seconds: water surface + Cornell box,
    200 000 emitted photons, k = 500, 960x540 -- every diffuse pixel is a gather."""
    import os
    rgbs, ior = (0.26, 0.84, 1.0, 1.8), 1.33
    s = build_scene(rt.Scene(), "water.obj", rgbs, ior)
    orc = build_scene(Oracle(), "water.obj", rgbs, ior)
    s.emit_photons(200000, seed=12345)
    ph = s.photons_get()
    orc.photons_set(np.c_[ph, np.zeros(len(ph))])
    orc.build_kdtree()
    gpu_ctx.upload(s)
    gpu_ctx.photons_upload(ph)
    w, h = 960, 540
    p = rt.default_params(width=w, height=h, photon_num=200000, photon_use=500)
    rgb, bgr, st = gpu_ctx.render(rt.look_at(CAM[:3], CAM[3:]), p)
    orc.params.head_quirk = 0
    orc.params.photon_num = 200000
    orc.params.photon_use = 500
    ref, _ = orc.render(w, h, CAM, threads=min(16, os.cpu_count() or 1))
    assert_image_close(rgb, ref, what="C4 960x540")
    assert st.gather_queries > 100000


def test_full_size_photon_map_frame(gpu_ctx, monkeypatch):
    """Configuration C4 at its full size: water surface + Cornell box, 1 M photons emitted, ordered and indexed on the
    device, k = 1000, 1920x1080.  Two checks that do not need the oracle to render 1.3 M gathers: the grouped gather
    and the one-request-per-wave gather (different algorithms, same k-nearest sets) give the same frame, and a band
    of rows through the middle of the frame matches the oracle given the same photon list."""
    import os
    rgbs, ior = (0.26, 0.84, 1.0, 1.8), 1.33
    s = build_scene(rt.Scene(), "water.obj", rgbs, ior)
    orc = build_scene(Oracle(), "water.obj", rgbs, ior)
    gpu_ctx.upload(s)
    n_emit = 1000000
    (n_surface, n_volume), _ = gpu_ctx.emit_photon_maps(n_emit, seed=12345)
    assert n_surface > 1000000 and n_volume == 0
    w, h = 1920, 1080
    cam = rt.look_at(CAM[:3], CAM[3:])
    p = rt.default_params(width=w, height=h, photon_num=n_emit, photon_use=1000)
    rgb, bgr, st = gpu_ctx.render(cam, p)
    assert st.gather_queries > 1000000 and st.gather_single < st.gather_queries // 4      # most gathers went through groups
    monkeypatch.setenv("RT_GATHER_MODE", "query")
    rgb_q, bgr_q, st_q = gpu_ctx.render(cam, p)
    monkeypatch.delenv("RT_GATHER_MODE")
    assert st_q.gather_single == st_q.gather_queries == st.gather_queries
    err = np.abs(rgb - rgb_q).max(axis=-1)
    # (the same sets -- photons at exactly the k-th distance are chosen by one rule -- summed in another order, the grouped
    # form four float terms at a time)
    assert (err < 1e-7).mean() >= 0.999 and err.max() < 1e-6 and (bgr != bgr_q).mean() < 1e-5, ((err < 1e-7).mean(), err.max(), (bgr != bgr_q).mean())
    ph = gpu_ctx.photon_map_get(0)
    assert ph.shape == (n_surface, 9)
    orc.photons_set(np.c_[ph, np.zeros(len(ph))])
    orc.build_kdtree()
    orc.params.head_quirk = 0
    orc.params.photon_num = n_emit
    orc.params.photon_use = 1000
    r0, r1 = 536, 544
    ref, _ = orc.render(w, h, CAM, rows=(r0, r1), threads=min(16, os.cpu_count() or 1))
    assert_image_close(rgb[r0:r1], ref[r0:r1], what="C4 1920x1080 rows 536..543")
    assert ref[r0:r1].max() > 0.05
