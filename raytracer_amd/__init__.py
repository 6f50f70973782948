"""raytracer_amd: MI355X (gfx950) ray-trace hot path behind the reference renderer's scene API.

The product is the C-ABI library librt_amd.so (HIP kernels + host scene code, sources in csrc/)
and the C++ scene headers in scene/.  This Python package is thin plumbing over that ABI for the
tests and the benchmark: numpy arrays in, numpy arrays out, torch device pointers passed through.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import RT_F32, RT_F64, Camera, RenderParams, Stats, lib

__all__ = ["Scene", "Context", "RtError", "RT_F32", "RT_F64", "default_params", "look_at", "save_bmp", "gather_rows", "aligned_bytes"]


class RtError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__(f"{where}: status {code}: {lib.rt_last_error().decode(errors='replace')}")


def _chk(code, where):
    if code < 0:
        raise RtError(code, where)
    return code


def _dp(a):
    return a.ctypes.data_as(_lib.P(_lib.c_d))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def default_params(**kw):
    p = RenderParams()
    lib.rt_default_params(ctypes.byref(p))
    for k, v in kw.items():
        if k == "light_pos":
            for i in range(3):
                p.light_pos[i] = v[i]
        else:
            setattr(p, k, v)
    return p


def look_at(pos=(2.8, 0, 0), target=(0, 0, 0)):
    cam = Camera()
    lib.rt_camera_look_at(_dp(_f64(pos)), _dp(_f64(target)), ctypes.byref(cam))
    return cam


def aligned_bytes(n, align=64):
    """A writable uint8 array of n bytes whose first byte sits at a multiple of `align` (flat scene images need 64)."""
    raw = np.empty(n + align, np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n]


def _as_image(buf):
    """uint8 view of a flat scene image at a 64-byte aligned address (copied only when it is not there already)."""
    a = np.ascontiguousarray(np.asarray(buf).view(np.uint8).ravel())
    if a.ctypes.data % 64:
        b = aligned_bytes(a.size)
        b[:] = a
        a = b
    return a


class Scene:
    """Host scene assembly (mirrors `vector<Object*> scene_objects` + `new BVH(scene_objects)`)."""

    def __init__(self):
        h = ctypes.c_void_p()
        _chk(lib.rt_scene_create(ctypes.byref(h)), "rt_scene_create")
        self.h = h

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:      # (module globals are gone at interpreter shutdown)
            lib.rt_scene_destroy(self.h)
            self.h = None

    def size(self):
        return lib.rt_scene_size(self.h)

    def add_triangle(self, A, C, B, rgbs, ior=1.0):
        return _chk(lib.rt_scene_add_triangle(self.h, _dp(_f64(A)), _dp(_f64(C)), _dp(_f64(B)), _dp(_f64(rgbs)), ior), "rt_scene_add_triangle")

    def set_normals(self, obj, na, nc, nb):
        _chk(lib.rt_scene_set_normals(self.h, obj, _dp(_f64(na)), _dp(_f64(nc)), _dp(_f64(nb))), "rt_scene_set_normals")

    def set_texture(self, obj, ta, tc, tb, use, map_idx):
        _chk(lib.rt_scene_set_texture(self.h, obj, _dp(_f64(ta)), _dp(_f64(tc)), _dp(_f64(tb)), int(use), map_idx), "rt_scene_set_texture")

    def add_sphere(self, c, radius, rgbs, ior=1.0):
        return _chk(lib.rt_scene_add_sphere(self.h, _dp(_f64(c)), radius, _dp(_f64(rgbs)), ior), "rt_scene_add_sphere")

    def add_plane(self, normal, distance, rgbs):
        return _chk(lib.rt_scene_add_plane(self.h, _dp(_f64(normal)), distance, _dp(_f64(rgbs))), "rt_scene_add_plane")

    def add_cornell(self, c1=(1, 1, 1), c2=(-1, -1, -1)):
        _chk(lib.rt_scene_add_cornell_box(self.h, _dp(_f64(c1)), _dp(_f64(c2))), "rt_scene_add_cornell_box")

    def add_cube(self, c1, c2, rgbs):
        _chk(lib.rt_scene_add_cube(self.h, _dp(_f64(c1)), _dp(_f64(c2)), _dp(_f64(rgbs))), "rt_scene_add_cube")

    def load_obj(self, path, rgbs=(0.5, 0.5, 0.5, 0.0), ior=1.0, off=(0, 0, 0), use_texture=False, map_idx=-1):
        _chk(lib.rt_scene_load_obj(self.h, path.encode(), _dp(_f64(rgbs)), ior, off[0], off[1], off[2], int(use_texture), map_idx),
             "rt_scene_load_obj")
        return 1

    def add_texture_bmp(self, path):
        return _chk(lib.rt_scene_add_texture_bmp(self.h, path.encode()), "rt_scene_add_texture_bmp")

    def add_texture_rgb(self, rgb):
        rgb = _f64(rgb)
        return _chk(lib.rt_scene_add_texture_rgb(self.h, rgb.shape[0], rgb.shape[1], _dp(rgb)), "rt_scene_add_texture_rgb")

    def build_bvh(self):
        _chk(lib.rt_scene_build_bvh(self.h), "rt_scene_build_bvh")

    def emit_photons(self, n_emit, light_pos=(0.1, 0.2, 0.0), seed=12345, bounce=5, accuracy=1e-6):
        """Host-side photon emission (main.cpp:697-718 + photonEmission); returns the surface photon count."""
        _chk(lib.rt_scene_emit_photons(self.h, n_emit, _dp(_f64(light_pos)), seed, bounce, accuracy), "rt_scene_emit_photons")
        return self.photon_count(0)

    def emit_photons_gpu(self, ctx, n_emit, light_pos=(0.1, 0.2, 0.0), seed=12345, bounce=5, accuracy=1e-6):
        """The same emission on the GPU (per-photon random streams); the scene must be the one uploaded to ctx."""
        st = Stats()
        _chk(lib.rt_scene_emit_photons_gpu(ctx.h, self.h, n_emit, _dp(_f64(light_pos)), seed, bounce, accuracy, ctypes.byref(st)),
             "rt_scene_emit_photons_gpu")
        return st

    def photon_count(self, which=0):
        return lib.rt_scene_photon_count(self.h, which)

    def photons_get(self, which=0):
        out = np.zeros((self.photon_count(which), 9))
        if len(out):
            _chk(lib.rt_scene_photons_get(self.h, which, _dp(out)), "rt_scene_photons_get")
        return out

    def photons_clear(self):
        lib.rt_scene_photons_clear(self.h)

    def bvh_depth(self):
        return lib.rt_scene_bvh_depth(self.h)

    def export_flat(self):
        """The built scene as one block of bytes (uint8 array, 64-byte aligned): rt_scene_export_flat."""
        n = ctypes.c_size_t(0)
        _chk(lib.rt_scene_export_flat(self.h, None, ctypes.byref(n)), "rt_scene_export_flat")
        buf = aligned_bytes(n.value)
        _chk(lib.rt_scene_export_flat(self.h, buf.ctypes.data, ctypes.byref(n)), "rt_scene_export_flat")
        return buf

    def import_flat(self, image):
        """Replace this scene by the one in a flat image: rt_scene_import_flat."""
        a = _as_image(image)
        _chk(lib.rt_scene_import_flat(self.h, a.ctypes.data, a.size), "rt_scene_import_flat")

    def bvh_dump(self):
        n_rec = ctypes.c_int64(0)
        n_box = ctypes.c_int64(0)
        _chk(lib.rt_scene_bvh_dump(self.h, None, ctypes.byref(n_rec), None, ctypes.byref(n_box)), "rt_scene_bvh_dump")
        rec = np.zeros(n_rec.value, np.int32)
        boxes = np.zeros((n_box.value, 6))
        _chk(lib.rt_scene_bvh_dump(self.h, rec.ctypes.data_as(_lib.P(_lib.c_i32)), ctypes.byref(n_rec), _dp(boxes), ctypes.byref(n_box)),
             "rt_scene_bvh_dump")
        return rec, boxes


class Context:
    """One GPU.  Raises RtError (RT_ERR_NO_DEVICE) when no gfx950 device is usable -- never falls back."""

    def __init__(self, device=0):
        h = ctypes.c_void_p()
        _chk(lib.rt_context_create(device, ctypes.byref(h)), "rt_context_create")
        self.h = h
        self.device = device

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib.rt_context_destroy(self.h)
            self.h = None

    @property
    def stream(self):
        return lib.rt_context_stream(self.h)

    def synchronize(self):
        _chk(lib.rt_context_synchronize(self.h), "rt_context_synchronize")

    def upload(self, scene):
        _chk(lib.rt_scene_upload(self.h, scene.h), "rt_scene_upload")

    def upload_flat(self, image):
        """rt_scene_upload from a flat image (Scene.export_flat of this or another process)."""
        a = _as_image(image)
        _chk(lib.rt_scene_upload_flat(self.h, a.ctypes.data, a.size), "rt_scene_upload_flat")

    def upload_lbvh(self, scene):
        """Upload with the tree built on the GPU (linear BVH, not the reference's tree); returns the build's device ms."""
        ms = ctypes.c_double(0)
        _chk(lib.rt_scene_upload_lbvh(self.h, scene.h, ctypes.byref(ms)), "rt_scene_upload_lbvh")
        return ms.value

    def trace_closest(self, org, d, max_dist=1e9, precision=RT_F64):
        org = _f64(org); d = _f64(d)
        n = org.shape[0]
        dist = np.full(n, max_dist, dtype=np.float64) if np.isscalar(max_dist) else _f64(max_dist).copy()
        ids = np.zeros(n, np.int32)
        _chk(lib.rt_trace_closest(self.h, precision, n, _dp(org), _dp(d), _dp(dist), ids.ctypes.data_as(_lib.P(_lib.c_i32))), "rt_trace_closest")
        return ids, dist

    def trace_closest_dev(self, n, d_org, d_dir, d_dist, d_ids, precision=RT_F64):
        """Device pointers (ints): SoA rays.  Enqueued on the context stream."""
        _chk(lib.rt_trace_closest_dev(self.h, precision, n, d_org, d_dir, d_dist, d_ids), "rt_trace_closest_dev")

    def trace_primary(self, cam, params, rows=None, want_t=True):
        part_rows = params.height // params.n_parts if params.band_rows > 0 else params.height
        r0, r1 = rows if rows else (0, part_rows)
        n = params.width * (r1 - r0)
        ids = np.zeros(n, np.int32)
        t = np.zeros(n) if want_t else None
        sh = np.zeros(n, np.int32) if params.shadow_rays else None
        st = Stats()
        _chk(lib.rt_trace_primary(self.h, ctypes.byref(cam), ctypes.byref(params), r0, r1, ids.ctypes.data_as(_lib.P(_lib.c_i32)),
                                  _dp(t) if want_t else None, sh.ctypes.data_as(_lib.P(_lib.c_i32)) if sh is not None else None,
                                  ctypes.byref(st)), "rt_trace_primary")
        return ids, t, sh, st

    def trace_primary_dev(self, cam, params, rows, d_ids, d_t=None, d_shadow=None, stats=True):
        st = Stats()
        _chk(lib.rt_trace_primary_dev(self.h, ctypes.byref(cam), ctypes.byref(params), rows[0], rows[1], d_ids, d_t, d_shadow,
                                      ctypes.byref(st) if stats else None), "rt_trace_primary_dev")
        return st


def _ctx_photons_upload(self, photons, which=0):
    photons = _f64(photons)
    assert photons.ndim == 2 and photons.shape[1] == 9
    _chk(lib.rt_photons_upload(self.h, which, photons.shape[0], _dp(photons)), "rt_photons_upload")


def _ctx_photons_upload_dev_build(self, photons, which=0):
    """As photons_upload, the kd-tree built on the GPU."""
    photons = _f64(photons)
    assert photons.ndim == 2 and photons.shape[1] == 9
    _chk(lib.rt_photons_upload_dev_build(self.h, which, photons.shape[0], _dp(photons)), "rt_photons_upload_dev_build")


def _ctx_emit_photon_maps(self, n_emit, light_pos=(0.1, 0.2, 0.0), seed=12345, bounce=5, accuracy=1e-6):
    """Emission + ordering + both kd-trees on the GPU; returns ((surface, volume) counts, Stats)."""
    st = Stats()
    counts = (_lib.c_i64 * 2)(0, 0)
    _chk(lib.rt_emit_photon_maps_dev(self.h, n_emit, _dp(_f64(light_pos)), seed, bounce, accuracy, counts, ctypes.byref(st)),
         "rt_emit_photon_maps_dev")
    return (counts[0], counts[1]), st


def _ctx_photon_map_get(self, which=0):
    n = lib.rt_photon_map_size(self.h, which)
    out = np.zeros((max(n, 0), 9))
    _chk(lib.rt_photon_map_get(self.h, which, _dp(out)), "rt_photon_map_get")
    return out


def _ctx_knn_query(self, k, centers, max_dist=0.28, which=0):
    """KDTree::findKNN for a batch: (idx [n, k] into the uploaded photon list, -1 padded, unordered; counts [n])."""
    centers = _f64(centers)
    n = centers.shape[0]
    idx = np.full((n, k), -1, np.int32)
    cnt = np.zeros(n, np.int32)
    _chk(lib.rt_knn_query(self.h, which, k, max_dist, n, _dp(centers), idx.ctypes.data_as(_lib.P(_lib.c_i32)),
                          cnt.ctypes.data_as(_lib.P(_lib.c_i32))), "rt_knn_query")
    return idx, cnt


def _ctx_knn_radius(self, k, centers, max_dist=0.28, which=0):
    centers = _f64(centers)
    n = centers.shape[0]
    cnt = np.zeros(n, np.int32)
    r2 = np.zeros(n)
    _chk(lib.rt_knn_radius(self.h, which, k, max_dist, n, _dp(centers), cnt.ctypes.data_as(_lib.P(_lib.c_i32)), _dp(r2)), "rt_knn_radius")
    return cnt, r2


def _ctx_knn_estimate(self, params, pos, normal, color):
    pos = _f64(pos); normal = _f64(normal); color = _f64(color)
    n = pos.shape[0]
    rgb = np.zeros((n, 3))
    cnt = np.zeros(n, np.int32)
    _chk(lib.rt_knn_estimate(self.h, ctypes.byref(params), n, _dp(pos), _dp(normal), _dp(color), _dp(rgb),
                             cnt.ctypes.data_as(_lib.P(_lib.c_i32))), "rt_knn_estimate")
    return rgb, cnt


def _ctx_knn_estimate_grouped(self, params, pos, normal, color):
    """knn_estimate through the grouped gather of the pixel path (64 consecutive requests per wave)."""
    pos = _f64(pos); normal = _f64(normal); color = _f64(color)
    n = pos.shape[0]
    rgb = np.zeros((n, 3))
    _chk(lib.rt_knn_estimate_grouped(self.h, ctypes.byref(params), n, _dp(pos), _dp(normal), _dp(color), _dp(rgb)), "rt_knn_estimate_grouped")
    return rgb


def _ctx_render(self, cam, params, rows=None):
    """Full pixel path for a row block: returns (rgb float64 [rows, w, 3], bgr8 uint8 [rows, w, 3], Stats)."""
    part_rows = params.height // params.n_parts if params.band_rows > 0 else params.height
    r0, r1 = rows if rows else (0, part_rows)
    n = params.width * (r1 - r0)
    rgb = np.zeros((r1 - r0, params.width, 3))
    bgr = np.zeros((r1 - r0, params.width, 3), np.uint8)
    st = Stats()
    _chk(lib.rt_render_rows(self.h, ctypes.byref(cam), ctypes.byref(params), r0, r1, _dp(rgb), bgr.ctypes.data_as(_lib.P(_lib.c_u8)),
                            ctypes.byref(st)), "rt_render_rows")
    return rgb, bgr, st


def _ctx_render_dev(self, cam, params, rows, d_rgb, d_bgr8, stats=True):
    st = Stats()
    _chk(lib.rt_render_rows_dev(self.h, ctypes.byref(cam), ctypes.byref(params), rows[0], rows[1], d_rgb, d_bgr8,
                                ctypes.byref(st) if stats else None), "rt_render_rows_dev")
    return st


Context.photons_upload = _ctx_photons_upload
Context.photons_upload_dev_build = _ctx_photons_upload_dev_build
Context.emit_photon_maps = _ctx_emit_photon_maps
Context.photon_map_get = _ctx_photon_map_get
Context.knn_query = _ctx_knn_query
Context.knn_radius = _ctx_knn_radius
Context.knn_estimate = _ctx_knn_estimate
Context.knn_estimate_grouped = _ctx_knn_estimate_grouped
Context.render = _ctx_render
Context.render_dev = _ctx_render_dev


def gather_rows(ctxs, d_parts, band_rows, height, row_bytes, d_frame):
    """rt_gather_rows: the compact band buffers d_parts[r] (device pointers, traced by ctxs[r]) -> the frame d_frame on
    the device of ctxs[0]."""
    n = len(ctxs)
    hs = (_lib.vp * n)(*[c.h for c in ctxs])
    ps = (_lib.vp * n)(*[ctypes.c_void_p(int(p) if not isinstance(p, ctypes.c_void_p) else p.value) for p in d_parts])
    _chk(lib.rt_gather_rows(hs, ps, n, band_rows, height, row_bytes, d_frame), "rt_gather_rows")


COMM_ID_BYTES = 128


def comm_unique_id():
    """rt_comm_unique_id: the 128-byte id rank 0 makes and hands to the other ranks (bytes)."""
    buf = (ctypes.c_uint8 * COMM_ID_BYTES)()
    _chk(lib.rt_comm_unique_id(buf), "rt_comm_unique_id")
    return bytes(buf)


class Comm:
    """The ranks of a job, one process and one GPU each, joined over RCCL through the C ABI (rt_comm_create: collective)."""

    def __init__(self, ctx, comm_id, rank, n_ranks):
        if len(comm_id) != COMM_ID_BYTES:
            raise ValueError("the id is %d bytes" % COMM_ID_BYTES)
        self.h = _lib.vp()
        self.ctx, self.rank, self.n_ranks = ctx, rank, n_ranks
        buf = (ctypes.c_uint8 * COMM_ID_BYTES).from_buffer_copy(comm_id)
        _chk(lib.rt_comm_create(ctx.h, buf, rank, n_ranks, ctypes.byref(self.h)), "rt_comm_create")

    def gather_frame(self, d_part, band_rows, height, row_bytes, d_frame=None):
        """rt_gather_frame (collective): this rank's compact bands d_part -> the frame d_frame on rank 0 (device pointers)."""
        _chk(lib.rt_gather_frame(self.h, d_part, band_rows, height, row_bytes, d_frame), "rt_gather_frame")

    def close(self):
        if self.h:
            lib.rt_comm_destroy(self.h)
            self.h = _lib.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def save_bmp(path, rgb, dpi=72):
    rgb = _f64(rgb)
    h, w = rgb.shape[0], rgb.shape[1]
    _chk(lib.rt_save_bmp(path.encode(), w, h, dpi, _dp(rgb)), "rt_save_bmp")
