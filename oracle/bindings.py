"""TEST INFRASTRUCTURE ONLY: ctypes bindings of the CPU oracle (librt_oracle.so) and, where the
reference is mounted, of the compiled reference (oracle/_ref/librt_ref.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.  Both
classes expose the same method names so that a check written once runs against either.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "librt_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "librt_ref.so")

c_d = ctypes.c_double
c_f = ctypes.c_float
c_i = ctypes.c_int32
c_l = ctypes.c_int64
P = ctypes.POINTER


def _dp(a):
    return a.ctypes.data_as(P(c_d))


def _ip(a):
    return a.ctypes.data_as(P(c_i))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Params(ctypes.Structure):
    _fields_ = [
        ("photon_num", c_i), ("photon_use", c_i), ("tracing_depth", c_i), ("bounce", c_i),
        ("head_quirk", c_i), ("use_photon_map", c_i), ("accuracy", c_d), ("photon_dist", c_d),
        ("focal_len", c_d), ("aperture", c_d), ("dof_samples", c_i), ("counter_rng", c_i), ("rng_seed", ctypes.c_uint32), ("pad_", c_i),
    ]


def have_reference():
    return os.path.exists(REF_SO)


class _Common:
    """Shared numpy-facing wrappers; subclasses provide self._call(name) -> function and self._h."""

    def primary_rays(self, w, h, cam):
        org = np.zeros((w * h, 3)); d = np.zeros((w * h, 3))
        self._fn("primary_rays")(c_i(w), c_i(h), _dp(_f64(cam)), _dp(org), _dp(d))
        return org, d

    def shoot(self, org, d, max_dist=1e9):
        org = _f64(org); d = _f64(d)
        n = org.shape[0]
        dist = np.full(n, max_dist, dtype=np.float64) if np.isscalar(max_dist) else _f64(max_dist).copy()
        ids = np.zeros(n, np.int32)
        self._fn("shoot")(*self._s(), c_l(n), _dp(org), _dp(d), _dp(dist), _ip(ids))
        return ids, dist

    def obj_intersect(self, obj, org, d):
        org = _f64(org); d = _f64(d)
        t = np.zeros(org.shape[0])
        self._fn("obj_intersect")(*self._s(), c_i(obj), c_l(org.shape[0]), _dp(org), _dp(d), _dp(t))
        return t

    def bbox_intersect(self, box, org, d, best):
        org = _f64(org); d = _f64(d); best = _f64(best)
        t = np.zeros(org.shape[0])
        self._fn("bbox_intersect")(_dp(_f64(box)), c_l(org.shape[0]), _dp(org), _dp(d), _dp(best), _dp(t))
        return t

    def obj_bbox(self, obj):
        b = np.zeros(6)
        self._fn("obj_bbox")(*self._s(), c_i(obj), _dp(b))
        return b

    def obj_normal(self, obj, pos):
        obj = _i32(obj); pos = _f64(pos); out = np.zeros_like(pos)
        self._fn("obj_normal")(*self._s(), _ip(obj), c_l(len(obj)), _dp(pos), _dp(out))
        return out

    def obj_tangent(self, obj, pos):
        obj = _i32(obj); pos = _f64(pos); out = np.zeros_like(pos)
        self._fn("obj_tangent")(*self._s(), _ip(obj), c_l(len(obj)), _dp(pos), _dp(out))
        return out

    def obj_color(self, obj, pos):
        obj = _i32(obj); pos = _f64(pos); out = np.zeros((len(obj), 4))
        self._fn("obj_color")(*self._s(), _ip(obj), c_l(len(obj)), _dp(pos), _dp(out))
        return out

    def reflect_dir(self, nrm, d):
        nrm = _f64(nrm); d = _f64(d); out = np.zeros_like(d)
        self._fn("reflect_dir")(c_l(d.shape[0]), _dp(nrm), _dp(d), _dp(out))
        return out

    def texture_get(self, idx, uv):
        uv = _f64(uv); out = np.zeros((uv.shape[0], 3))
        self._fn("texture_get")(*self._s(), c_i(idx), c_l(uv.shape[0]), _dp(uv), _dp(out))
        return out

    def texture_res(self, idx):
        a = c_i(); b = c_i()
        self._fn("texture_res")(*self._s(), c_i(idx), ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def bvh_dump(self):
        nb = c_l()
        fn = self._fn("bvh_dump"); fn.restype = c_l
        n = fn(*self._s(), ctypes.byref(nb))
        rec = np.zeros(n, np.int32); boxes = np.zeros((nb.value, 6))
        self._fn("bvh_dump_copy")(*self._s(), _ip(rec), _dp(boxes))
        return rec, boxes

    def kd_dump(self):
        fn = self._fn("kd_dump"); fn.restype = c_l
        n = fn(*self._s())
        rec = np.zeros(n, np.int32)
        self._fn("kd_dump_copy")(*self._s(), _ip(rec))
        return rec

    def knn(self, k, centers, which=0):
        centers = _f64(centers); n = centers.shape[0]
        idx = np.zeros((n, k), np.int32); cnt = np.zeros(n, np.int32); d2 = np.zeros((n, k))
        self._fn("knn")(*self._s(), c_i(which), c_i(k), c_l(n), _dp(centers), _ip(idx), _ip(cnt), _dp(d2))
        return idx, cnt, d2


class Oracle(_Common):
    """The CPU restatement (travels with the repo)."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError("oracle/librt_oracle.so missing: run __graft_entry__.build() or make -C oracle")
        self.lib = ctypes.CDLL(ORACLE_SO)
        self.lib.orc_scene_new.restype = ctypes.c_void_p
        self.h = ctypes.c_void_p(self.lib.orc_scene_new())
        self.params = Params()
        self.lib.orc_default_params(ctypes.byref(self.params))

    def __del__(self):
        try:
            self.lib.orc_scene_free(self.h)
        except Exception:
            pass

    def _fn(self, name):
        return getattr(self.lib, "orc_" + name)

    def _s(self):
        return (self.h,)

    def size(self):
        return self.lib.orc_scene_size(self.h)

    def load_obj(self, path, rgbs=(0.5, 0.5, 0.5, 0.0), ior=1.0, off=(0, 0, 0), use_texture=False, map_idx=-1):
        return self.lib.orc_load_obj(self.h, path.encode(), _dp(_f64(rgbs)), c_f(ior), c_f(off[0]), c_f(off[1]), c_f(off[2]),
                                     c_i(int(use_texture)), c_i(map_idx))

    def add_cornell(self, c1=(1, 1, 1), c2=(-1, -1, -1)):
        self.lib.orc_add_cornell(self.h, _dp(_f64(c1)), _dp(_f64(c2)))

    def add_cube(self, c1, c2, rgbs):
        self.lib.orc_add_cube(self.h, _dp(_f64(c1)), _dp(_f64(c2)), _dp(_f64(rgbs)))

    def add_triangle(self, A, C, B, rgbs, ior=1.0):
        return self.lib.orc_add_triangle(self.h, _dp(_f64(A)), _dp(_f64(C)), _dp(_f64(B)), _dp(_f64(rgbs)), c_f(ior))

    def last_set_normals(self, na, nc, nb):
        self.lib.orc_last_set_normals(self.h, _dp(_f64(na)), _dp(_f64(nc)), _dp(_f64(nb)))

    def last_set_texture(self, ta, tc, tb, use, map_idx):
        self.lib.orc_last_set_texture(self.h, _dp(_f64(ta)), _dp(_f64(tc)), _dp(_f64(tb)), c_i(int(use)), c_i(map_idx))

    def add_sphere(self, c, radius, rgbs, ior=1.0):
        return self.lib.orc_add_sphere(self.h, _dp(_f64(c)), c_d(radius), _dp(_f64(rgbs)), c_f(ior))

    def add_plane(self, normal, distance, rgbs):
        return self.lib.orc_add_plane(self.h, _dp(_f64(normal)), c_d(distance), _dp(_f64(rgbs)))

    def texture_read(self, path):
        return self.lib.orc_texture_read(self.h, path.encode())

    def build_bvh(self):
        self.lib.orc_build_bvh(self.h)

    def shoot_mt(self, org, d, max_dist=1e9, threads=8):
        org = _f64(org); d = _f64(d); n = org.shape[0]
        dist = np.full(n, max_dist, dtype=np.float64) if np.isscalar(max_dist) else _f64(max_dist).copy()
        ids = np.zeros(n, np.int32)
        self.lib.orc_shoot_mt(self.h, c_l(n), _dp(org), _dp(d), _dp(dist), _ip(ids), c_i(threads))
        return ids, dist

    def shoot_f32(self, org, d, max_dist=1e9, threads=8):
        org = _f64(org); d = _f64(d); n = org.shape[0]
        dist = np.full(n, max_dist, dtype=np.float64) if np.isscalar(max_dist) else _f64(max_dist).copy()
        ids = np.zeros(n, np.int32)
        self.lib.orc_shoot_f32(self.h, c_l(n), _dp(org), _dp(d), _dp(dist), _ip(ids), c_i(threads))
        return ids, dist

    def shoot_stats(self, org, d, max_dist=1e9):
        org = _f64(org); d = _f64(d)
        c = np.zeros(4, np.int64)
        self.lib.orc_shoot_stats(self.h, c_l(org.shape[0]), _dp(org), _dp(d), c_d(max_dist), c.ctypes.data_as(P(c_l)))
        return dict(slab_calls=int(c[0]), inner_visits=int(c[1]), prim_tests=int(c[2]), nodes_entered=int(c[3]))

    def brute_force(self, org, d, max_dist=1e9):
        org = _f64(org); d = _f64(d); n = org.shape[0]
        dist = np.full(n, max_dist, dtype=np.float64)
        ids = np.zeros(n, np.int32)
        self.lib.orc_brute_force(self.h, c_l(n), _dp(org), _dp(d), _dp(dist), _ip(ids))
        return ids, dist

    def primary_rays_aa(self, w, h, aadepth, cam):
        org = np.zeros((w * h * aadepth, 3)); d = np.zeros((w * h * aadepth, 3))
        self.lib.orc_primary_rays_aa(c_i(w), c_i(h), c_i(aadepth), _dp(_f64(cam)), _dp(org), _dp(d))
        return org, d

    def seed(self, s):
        self.lib.orc_seed(self.h, ctypes.c_uint32(s))

    def emit_photons(self, n, light_pos=(0.1, 0.2, 0.0)):
        self.lib.orc_emit_photons.restype = c_l
        return self.lib.orc_emit_photons(self.h, c_l(n), _dp(_f64(light_pos)), ctypes.byref(self.params))

    def photon_count(self, which=0):
        self.lib.orc_photon_count.restype = c_l
        return self.lib.orc_photon_count(self.h, c_i(which))

    def photons_get(self, which=0):
        out = np.zeros((self.photon_count(which), 10))
        self.lib.orc_photons_get(self.h, c_i(which), _dp(out))
        return out

    def photons_set(self, arr, which=0):
        arr = _f64(arr)
        self.lib.orc_photons_set(self.h, c_i(which), c_l(arr.shape[0]), _dp(arr))

    def build_kdtree(self):
        self.lib.orc_build_kdtree(self.h)

    def get_color_at(self, pos, d, obj, depth=1):
        pos = _f64(pos); d = _f64(d); obj = _i32(obj); out = np.zeros((len(obj), 3))
        self.lib.orc_get_color_at(self.h, ctypes.byref(self.params), c_l(len(obj)), _dp(pos), _dp(d), _ip(obj), c_i(depth), _dp(out))
        return out

    def render(self, w, h, cam, aadepth=1, rows=None, threads=8):
        rgb = np.zeros((h, w, 3))
        r0, r1 = rows if rows else (0, h)
        cnt = c_l()
        self.lib.orc_render(self.h, ctypes.byref(self.params), c_i(w), c_i(h), c_i(aadepth), _dp(_f64(cam)), c_i(r0), c_i(r1),
                            c_i(threads), _dp(rgb), ctypes.byref(cnt))
        return rgb, cnt.value

    def quantize(self, rgb):
        rgb = _f64(rgb).reshape(-1, 3)
        out = np.zeros((rgb.shape[0], 3), np.uint8)
        self.lib.orc_quantize(c_l(rgb.shape[0]), _dp(rgb), out.ctypes.data_as(P(ctypes.c_uint8)))
        return out

    def save_bmp(self, path, w, h, rgb, dpi=72):
        return self.lib.orc_save_bmp(path.encode(), c_i(w), c_i(h), c_i(dpi), _dp(_f64(rgb)))


class Reference(_Common):
    """The compiled, untouched reference (only where /root/reference is mounted).  Global state:
    one scene per process."""

    def __init__(self):
        if not have_reference():
            raise RuntimeError("oracle/_ref/librt_ref.so not built (reference not mounted?)")
        self.lib = ctypes.CDLL(REF_SO)
        self.lib.ref_scene_reset()
        self.n_maps = 0

    def _fn(self, name):
        return getattr(self.lib, "ref_" + name)

    def _s(self):
        return ()

    def size(self):
        return self.lib.ref_scene_size()

    def load_obj(self, path, rgbs=(0.5, 0.5, 0.5, 0.0), ior=1.0, off=(0, 0, 0), use_texture=False, tex_path=None):
        self.n_maps += 1
        return self.lib.ref_scene_load_obj(path.encode(), c_d(rgbs[0]), c_d(rgbs[1]), c_d(rgbs[2]), c_d(rgbs[3]), c_f(ior),
                                           c_f(off[0]), c_f(off[1]), c_f(off[2]), ctypes.c_int(int(use_texture)),
                                           tex_path.encode() if tex_path else None)

    def add_cornell(self, c1=(1, 1, 1), c2=(-1, -1, -1)):
        self.lib.ref_scene_add_cornell(_dp(_f64(c1)), _dp(_f64(c2)))

    def add_cube(self, c1, c2, rgbs):
        self.lib.ref_scene_add_cube(_dp(_f64(c1)), _dp(_f64(c2)), _dp(_f64(rgbs)))

    def add_triangle(self, A, C, B, rgbs, ior=1.0):
        self.lib.ref_scene_add_triangle(_dp(_f64(A)), _dp(_f64(C)), _dp(_f64(B)), _dp(_f64(rgbs)), c_f(ior))

    def last_set_normals(self, na, nc, nb):
        self.lib.ref_scene_last_set_normals(_dp(_f64(na)), _dp(_f64(nc)), _dp(_f64(nb)))

    def last_set_texture(self, ta, tc, tb, use, map_idx):
        self.lib.ref_scene_last_set_texture(_dp(_f64(ta)), _dp(_f64(tc)), _dp(_f64(tb)), ctypes.c_int(int(use)), ctypes.c_int(map_idx))

    def add_sphere(self, c, radius, rgbs, ior=1.0):
        self.lib.ref_scene_add_sphere(_dp(_f64(c)), c_d(radius), _dp(_f64(rgbs)), c_f(ior))

    def texture_read(self, path):
        self.n_maps += 1
        return self.lib.ref_texture_read(path.encode())

    def build_bvh(self):
        self.lib.ref_build_bvh()

    def seed(self, s):
        self.lib.ref_seed(ctypes.c_uint(s))

    def emit_photons(self, n, light_pos=(0.1, 0.2, 0.0)):
        self.lib.ref_emit_photons.restype = c_l
        return self.lib.ref_emit_photons(c_l(n), _dp(_f64(light_pos)))

    def photon_count(self, which=0):
        if which == 0:
            # size is only returned by emit; query through a zero-emission call
            self.lib.ref_emit_photons.restype = c_l
            return self.lib.ref_emit_photons(c_l(0), _dp(_f64((0, 0, 0))))
        self.lib.ref_volume_photon_count.restype = c_l
        return self.lib.ref_volume_photon_count()

    def photons_get(self, which=0):
        out = np.zeros((self.photon_count(which), 10))
        self.lib.ref_photons_get(ctypes.c_int(which), _dp(out))
        return out

    def photons_set(self, arr, which=0):
        arr = _f64(arr)
        self.lib.ref_photons_set(ctypes.c_int(which), c_l(arr.shape[0]), _dp(arr))

    def build_kdtree(self):
        self.lib.ref_build_kdtree()

    def get_color_at(self, pos, d, obj, depth=1):
        pos = _f64(pos); d = _f64(d); obj = _i32(obj); out = np.zeros((len(obj), 3))
        self.lib.ref_get_color_at(c_l(len(obj)), _dp(pos), _dp(d), _ip(obj), ctypes.c_int(depth), _dp(out))
        return out

    def save_bmp(self, path, w, h, rgb, dpi=72):
        self.lib.ref_save_bmp(path.encode(), ctypes.c_int(w), ctypes.c_int(h), ctypes.c_int(dpi), _dp(_f64(rgb)))

    def color_intended(self, pos, d, obj, depth=1):
        """The shade loop recursing on the newly hit object, driven through the reference's own functions."""
        pos = _f64(pos); d = _f64(d); obj = _i32(obj); out = np.zeros((len(obj), 3))
        self.lib.ref_color_intended(c_l(len(obj)), _dp(pos), _dp(d), _ip(obj), ctypes.c_int(depth), _dp(out))
        return out

    def primary_rays_aa(self, w, h, aadepth, cam):
        org = np.zeros((w * h * aadepth, 3)); d = np.zeros((w * h * aadepth, 3))
        self.lib.ref_primary_rays_aa(ctypes.c_int(w), ctypes.c_int(h), ctypes.c_int(aadepth), _dp(_f64(cam)), _dp(org), _dp(d))
        return org, d

    def render(self, w, h, cam, aadepth=1, intended=True):
        rgb = np.zeros((h, w, 3))
        self.lib.ref_render(ctypes.c_int(w), ctypes.c_int(h), ctypes.c_int(aadepth), _dp(_f64(cam)), ctypes.c_int(int(intended)), _dp(rgb))
        return rgb

    def load_obj_no_normals(self, path, rgbs=(0.5, 0.5, 0.5, 0.0), ior=1.0, off=(0, 0, 0), use_texture=False, map_idx=-1):
        """Meshes whose faces have no normal index (the reference's own reader crashes on them)."""
        return self.lib.ref_scene_load_obj_no_normals(path.encode(), c_d(rgbs[0]), c_d(rgbs[1]), c_d(rgbs[2]), c_d(rgbs[3]), c_f(ior),
                                                      c_f(off[0]), c_f(off[1]), c_f(off[2]), ctypes.c_int(int(use_texture)), ctypes.c_int(map_idx))


def id_map_hash(ids):
    """FNV-style hash of a hit-ID map: h = (h ^ uint32(id + 1)) * 1099511628211 mod 2^64."""
    ids = np.asarray(ids, dtype=np.int64)
    h = 1469598103934665603
    M = (1 << 64) - 1
    for v in ((ids + 1) & 0xFFFFFFFF).tolist():
        h = ((h ^ v) * 1099511628211) & M
    return "%016x" % h
