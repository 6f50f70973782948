"""The grouped photon gather of the pixel path (64 neighbouring requests per wave, one per lane)
against the per-request gather: same k-nearest sets, so the same estimates up to summation order."""
import numpy as np
import pytest

import raytracer_amd as rt
from conftest import build_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def water_photons(gpu_ctx):
    s = build_scene(rt.Scene(), "water.obj", (0.26, 0.84, 1.0, 1.8), 1.33, (0, 0, 0))
    s.emit_photons(60000, seed=7)
    return s.photons_get(0)


def _wall_queries(side, order="raster", seed=0):
    """Points just inside the five visible walls with their normals, in raster order per wall."""
    u, v = np.meshgrid(np.linspace(-0.97, 0.97, side), np.linspace(-0.97, 0.97, side), indexing="ij")
    pos, nrm = [], []
    for ax, val in [(1, -1.0), (1, 1.0), (2, -1.0), (2, 1.0), (0, -1.0)]:
        p = np.zeros((side * side, 3)); n = np.zeros((side * side, 3))
        others = [a for a in range(3) if a != ax]
        p[:, others[0]] = u.ravel(); p[:, others[1]] = v.ravel(); p[:, ax] = val * 0.999
        n[:, ax] = -val
        pos.append(p); nrm.append(n)
    pos, nrm = np.concatenate(pos), np.concatenate(nrm)
    if order == "shuffled":
        perm = np.random.default_rng(seed).permutation(len(pos))
        pos, nrm = pos[perm], nrm[perm]
    return pos, nrm


def _compare(ctx, p, pos, nrm, what):
    col = np.random.default_rng(3).uniform(0.2, 1.0, pos.shape)
    want, _ = ctx.knn_estimate(p, pos, nrm, col)
    got = ctx.knn_estimate_grouped(p, pos, nrm, col)
    err = np.abs(got - want).max(axis=1)
    # the two forms add the same float terms over the same sets (photons at exactly the k-th distance are chosen by one rule
    # in both), the grouped one four at a time in float before its double sum: float rounding apart
    assert (err < 1e-7).mean() >= 0.999, (what, (err < 1e-7).mean())
    assert err.max() < 1e-6 * max(1.0, want.max()), (what, err.max())
    assert want.max() > 0.05      # the comparison is not about zeros


@pytest.mark.parametrize("k", [1000, 100, 8])
def test_grouped_gather_matches_the_per_request_gather(gpu_ctx, water_photons, k):
    gpu_ctx.photons_upload(water_photons)
    p = rt.default_params(photon_num=60000, photon_use=k)
    pos, nrm = _wall_queries(40)
    _compare(gpu_ctx, p, pos, nrm, f"raster k={k}")
    _compare(gpu_ctx, p, pos[:1000 + 37], nrm[:1000 + 37], f"ragged last group k={k}")


def test_incoherent_groups_fall_back(gpu_ctx, water_photons):
    """Requests in random order: a group's box spans the room, the lists overflow or the bins crowd,
    and the requests go to the per-request kernel; the answers must not change."""
    gpu_ctx.photons_upload(water_photons)
    pos, nrm = _wall_queries(24, "shuffled")
    for k in (1000, 50):
        _compare(gpu_ctx, rt.default_params(photon_num=60000, photon_use=k), pos, nrm, f"shuffled k={k}")


def test_sparse_and_empty_maps(gpu_ctx, water_photons):
    pos, nrm = _wall_queries(16)
    p = rt.default_params(photon_num=60000, photon_use=1000)
    gpu_ctx.photons_upload(water_photons[:700])          # fewer than k photons anywhere: every set is "all within the radius"
    col = np.full(pos.shape, 0.8)
    want, cnt = gpu_ctx.knn_estimate(p, pos, nrm, col)
    got = gpu_ctx.knn_estimate_grouped(p, pos, nrm, col)
    assert cnt.max() < 1000 and cnt.max() > 0
    assert np.abs(got - want).max() < 1e-7
    gpu_ctx.photons_upload(water_photons[:0])
    assert np.array_equal(gpu_ctx.knn_estimate_grouped(p, pos, nrm, col), np.zeros_like(pos))


def test_duplicate_photons(gpu_ctx, water_photons):
    """Many photons at identical positions: equal distances crowd one bin, which the group form hands over."""
    ph = np.concatenate([water_photons[:3000]] * 40)
    gpu_ctx.photons_upload(ph)
    pos, nrm = _wall_queries(12)
    p = rt.default_params(photon_num=60000, photon_use=500)
    col = np.full(pos.shape, 0.8)
    want, _ = gpu_ctx.knn_estimate(p, pos, nrm, col)
    got = gpu_ctx.knn_estimate_grouped(p, pos, nrm, col)
    # with 40-fold duplicates the k-th distance is shared by many photons: which of them belong to the set is decided by
    # one rule (position in the photon tree) in the per-request kernel, which is where the group form sends such requests
    assert np.abs(got - want).max() < 1e-6 * max(want.max(), 1e-9)


def test_photon_pairs_a_rounding_error_apart(gpu_ctx, water_photons):
    """Every photon has a twin of another power at (in float) the same position, so the k-th and (k+1)-th distances
    coincide for about half the requests and only one of the two belongs.  The choice is by position in the photon tree
    in both forms -- the group form hands such requests on -- so they agree on every request, and a second run repeats the
    first to the last bit of the per-request form and to the order of the sums in the grouped one."""
    a = water_photons[:20000].copy()
    b = a.copy()
    b[:, :3] += 1e-9
    b[:, 3:6] *= 0.25
    ph = np.empty((2 * len(a), 9))
    ph[0::2], ph[1::2] = a, b
    gpu_ctx.photons_upload(ph)
    pos, nrm = _wall_queries(20)
    col = np.full(pos.shape, 0.8)
    for k in (101, 500):
        p = rt.default_params(photon_num=60000, photon_use=k)
        want, _ = gpu_ctx.knn_estimate(p, pos, nrm, col)
        got = gpu_ctx.knn_estimate_grouped(p, pos, nrm, col)
        scale = max(want.max(), 1e-9)
        assert want.max() > 0.01
        assert np.abs(got - want).max() < 1e-6 * scale, (k, np.abs(got - want).max() / scale)
        # shuffled: other neighbours before each request, other starting bounds, the same sets
        perm = np.random.default_rng(k).permutation(len(pos))
        again, _ = gpu_ctx.knn_estimate(p, pos[perm], nrm[perm], col[perm])
        assert np.abs(again - want[perm]).max() < 1e-12 * scale
        assert np.abs(gpu_ctx.knn_estimate_grouped(p, pos, nrm, col) - got).max() < 1e-12 * scale


def test_scheduling_of_the_group_kernel_does_not_change_results(gpu_ctx, water_photons, monkeypatch):
    """Which wave takes a window, or a part of one, and in which order the windows are drawn is scheduling: a round is
    computed from the window and the lanes waiting alone.  Windows of several patches (runs of 16 neighbours from different
    walls) with parts handed from wave to wave, without, in queue order, with rounds of any size, and with short packed
    lists (more repeated searches, more boxes too large): the same estimates to the last bits.  The earlier group form over
    listed leaves adds in another order: float rounding apart."""
    gpu_ctx.photons_upload(water_photons)
    pos, nrm = _wall_queries(40)
    n = len(pos) // 16 * 16
    perm = np.random.default_rng(11).permutation(n // 16)
    idx = (perm[:, None] * 16 + np.arange(16)[None, :]).ravel()          # runs of 16 neighbours, the runs shuffled
    pos, nrm = pos[idx], nrm[idx]
    col = np.random.default_rng(3).uniform(0.2, 1.0, pos.shape)
    for k in (1000, 100):
        p = rt.default_params(photon_num=60000, photon_use=k)
        base = gpu_ctx.knn_estimate_grouped(p, pos, nrm, col)
        want, _ = gpu_ctx.knn_estimate(p, pos, nrm, col)
        scale = max(1.0, want.max())
        assert want.max() > 0.05 and np.abs(base - want).max() < 1e-6 * scale
        for env in ({"RT_GATHER_NO_PARTS": "1"}, {"RT_GATHER_QUEUE_ORDER": "1"}, {"RT_GATHER_PART_TAIL": "0"}, {"RT_GATHER_WAVES": "3"}):
            for name, val in env.items():
                monkeypatch.setenv(name, val)
            got = gpu_ctx.knn_estimate_grouped(p, pos, nrm, col)
            for name in env:
                monkeypatch.delenv(name)
            assert np.abs(got - base).max() < 1e-12 * scale, (k, env, np.abs(got - base).max())
        # other groupings of the same requests: same sets, other sums
        for env in ({"RT_GATHER_MIN_LANES": "1"}, {"RT_GATHER_PACK_CAP": "2048"}, {"RT_GATHER_ACT_REACH": "1.0"}, {"RT_GATHER_FORM": "tile"}):
            for name, val in env.items():
                monkeypatch.setenv(name, val)
            got = gpu_ctx.knn_estimate_grouped(p, pos, nrm, col)
            for name in env:
                monkeypatch.delenv(name)
            assert np.abs(got - want).max() < 1e-6 * scale, (k, env, np.abs(got - want).max())
