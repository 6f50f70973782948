"""Photon maps built on the GPU (the kd-tree of Kdtree.cpp:11-52 level by level, and the whole
emission -> ordering -> trees pipeline of main.cpp:683-727 without a host round trip) against the
host-built map and against exhaustive search."""
import numpy as np
import pytest

import raytracer_amd as rt
from conftest import CAM, build_scene

pytestmark = pytest.mark.gpu


def _photons(n, kind, seed):
    rng = np.random.default_rng(seed)
    ph = np.zeros((n, 9))
    if kind == "box":
        ph[:, :3] = rng.uniform(-1, 1, (n, 3))
    elif kind == "walls":            # photons on the six faces of the box: one coordinate shared by many photons
        ph[:, :3] = rng.uniform(-1, 1, (n, 3))
        face = rng.integers(0, 6, n)
        ph[np.arange(n), face % 3] = np.where(face < 3, -1.0, 1.0)
    else:                            # clusters far narrower than the gather radius
        c = rng.uniform(-1, 1, (max(1, n // 500), 3))
        ph[:, :3] = c[rng.integers(0, len(c), n)] + rng.normal(0, 0.01, (n, 3))
    ph[:, 3:6] = rng.uniform(0, 1, (n, 3))
    d = rng.normal(size=(n, 3))
    ph[:, 6:9] = d / np.linalg.norm(d, axis=1, keepdims=True)
    return ph


def _sets(idx, cnt):
    return [frozenset(row[:c].tolist()) for row, c in zip(idx, cnt)]


@pytest.mark.parametrize("n", [0, 1, 50, 64, 65, 129, 1000, 4097, 300000])
@pytest.mark.parametrize("kind", ["box", "walls", "clusters"])
def test_device_built_map_answers_like_the_host_built_one(gpu_ctx, n, kind):
    ph = _photons(n, kind, seed=n + len(kind))
    rng = np.random.default_rng(1)
    q = np.concatenate([rng.uniform(-1, 1, (150, 3)), ph[rng.integers(0, max(n, 1), 50) % max(n, 1), :3] + 0.003 if n else np.zeros((0, 3))])
    for k, r in ((8, 0.28), (50, 0.28), (200, 0.1)):
        gpu_ctx.photons_upload(ph)
        hi, hc = gpu_ctx.knn_query(k, q, max_dist=r)
        _, hr2 = gpu_ctx.knn_radius(k, q, max_dist=r)
        gpu_ctx.photons_upload_dev_build(ph)
        di, dc = gpu_ctx.knn_query(k, q, max_dist=r)
        _, dr2 = gpu_ctx.knn_radius(k, q, max_dist=r)
        assert np.array_equal(hc, dc)
        assert np.array_equal(hr2, dr2)          # the k-th distance is the same number whatever the layout
        same = np.mean([a == b for a, b in zip(_sets(hi, hc), _sets(di, dc))])
        assert same >= 0.995, (n, kind, k, same)  # sets differ only where two photons tie in float distance at the k-th place
    if 0 < n <= 4097:
        # exhaustive search on the float positions the device holds, float arithmetic as the kernel's
        p32 = ph[:, :3].astype(np.float32)
        q32 = q.astype(np.float32)
        dd = q32[:, None, :] - p32[None, :, :]
        d2 = dd[..., 0] * dd[..., 0] + dd[..., 1] * dd[..., 1] + dd[..., 2] * dd[..., 2]
        k, r = 50, 0.28
        inside = d2 < np.float32(r * r)
        want = np.minimum(inside.sum(1), k)
        di, dc = gpu_ctx.knn_query(k, q, max_dist=r)
        assert np.array_equal(dc, want)
        for row, c, drow, ins in zip(di, dc, d2, inside):
            if c == 0:
                continue
            kth = np.sort(drow[ins])[c - 1]
            assert np.all(drow[row[:c]] <= kth)


def test_emission_to_maps_on_the_device(gpu_ctx):
    """Same photons as the emission that returns host lists; both maps usable by the render; two runs
    give the same image up to the order in which a pixel's contributions are added."""
    s = build_scene(rt.Scene(), "teapot_normal.obj", (0.9, 0.7, 0.6, -1.0), 1.3, (0, -0.7, 0))     # subsurface teapot: both maps fill
    gpu_ctx.upload(s)
    n = 200000
    s.emit_photons_gpu(gpu_ctx, n, seed=9)
    lists = [s.photons_get(0), s.photons_get(1)]
    (ns, nv), st = gpu_ctx.emit_photon_maps(n, seed=9)
    assert (ns, nv) == (len(lists[0]), len(lists[1])) and nv > 1000
    assert st.kernel_ms >= st.trace_kernel_ms > 0
    for which in (0, 1):
        assert np.array_equal(gpu_ctx.photon_map_get(which), lists[which])
    cam = rt.look_at(CAM[:3], CAM[3:])
    p = rt.default_params(width=160, height=120, photon_num=n)
    a, _, _ = gpu_ctx.render(cam, p)
    gpu_ctx.emit_photon_maps(n, seed=9)
    b, _, _ = gpu_ctx.render(cam, p)
    assert np.abs(a - b).max() < 1e-9, np.abs(a - b).max()
    gpu_ctx.photons_upload(lists[0], 0)
    gpu_ctx.photons_upload(lists[1], 1)
    c, _, _ = gpu_ctx.render(cam, p)
    # host-built trees hold the same photons in another order: sums differ in the last bits only
    assert np.abs(a - c).max() < 1e-4, np.abs(a - c).max()


def test_device_map_errors(gpu_ctx):
    gpu_ctx.photons_upload(_photons(100, "box", 0))
    with pytest.raises(rt.RtError):
        gpu_ctx.photon_map_get(0)            # a host-built map keeps no list on the device


def test_emission_bounce_limits(gpu_ctx):
    """BOUNCE (main.cpp:25) from 0 (the first hit only) up: one launch per bounce with the survivors handed on; both
    device routes store the same photons, a longer walk stores more of them, and the serial host walk of the same
    limit stores about as many."""
    n = 60000
    stored = []
    for bounce in (0, 1, 7):
        # (a fresh scene per limit: both emissions append to the scene's lists, as main.cpp:89 does)
        s = build_scene(rt.Scene(), "teapot_normal.obj", (0.9, 0.9, 0.9, 0.6), 1.3, (0, -0.7, 0))      # a mirror teapot in the box
        gpu_ctx.upload(s)
        s.emit_photons_gpu(gpu_ctx, n, seed=3, bounce=bounce)
        lst = s.photons_get(0)
        (ns, nv), _ = gpu_ctx.emit_photon_maps(n, seed=3, bounce=bounce)
        assert ns == len(lst) and nv == 0 and np.array_equal(gpu_ctx.photon_map_get(0), lst)
        s.emit_photons(n, seed=3, bounce=bounce)
        host = s.photon_count(0) - ns
        assert abs(ns - host) < 0.02 * host + 50, (bounce, ns, host)
        stored.append(ns)
    assert 0 < stored[0] <= n and stored[0] < stored[1] < stored[2]
    with pytest.raises(rt.RtError):
        gpu_ctx.emit_photon_maps(n, seed=3, bounce=8)        # (a photon's records are keyed photon * 64 + bounce * 8 + slot)
