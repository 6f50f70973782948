"""ctypes view of librt_amd.so (the C ABI declared in include/rt_amd.h).

Plumbing only: loads the in-tree shared library and declares the prototypes.  There is no
fallback of any kind -- if the library is missing, importing this module raises.

A process that also uses torch must import torch (and initialise its GPU runtime) BEFORE this
module: the torch wheel bundles its own HIP runtime, and the runtime loaded first is the one
that owns the devices.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librt_amd.so")

c_d = ctypes.c_double
c_f = ctypes.c_float
c_i = ctypes.c_int
c_i32 = ctypes.c_int32
c_i64 = ctypes.c_int64
c_u8 = ctypes.c_uint8
P = ctypes.POINTER
vp = ctypes.c_void_p


class Camera(ctypes.Structure):
    _fields_ = [("pos", c_d * 3), ("dir", c_d * 3), ("right", c_d * 3), ("down", c_d * 3)]


class RenderParams(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("width", c_i32), ("height", c_i32), ("aadepth", c_i32), ("tracing_depth", c_i32),
        ("photon_num", c_i32), ("photon_use", c_i32), ("head_quirk", c_i32), ("use_photon_map", c_i32), ("precision", c_i32),
        ("shadow_rays", c_i32), ("count_work", c_i32), ("band_rows", c_i32), ("n_parts", c_i32), ("part", c_i32), ("accuracy", c_d), ("photon_dist", c_d), ("light_pos", c_d * 3),
        ("focal_len", c_d), ("aperture", c_d), ("dof_samples", c_i32), ("rng_seed", ctypes.c_uint32),
    ]


class Stats(ctypes.Structure):
    _fields_ = [
        ("primary_rays", ctypes.c_uint64), ("secondary_rays", ctypes.c_uint64), ("shadow_rays", ctypes.c_uint64),
        ("gather_queries", ctypes.c_uint64), ("inner_visits", ctypes.c_uint64), ("prim_tests", ctypes.c_uint64), ("kernel_ms", c_d), ("trace_kernel_ms", c_d), ("gather_single", ctypes.c_uint64), ("tile_order", ctypes.c_uint64),
    ]


RT_F64, RT_F32 = 0, 1

# name -> (restype, argtypes); every symbol include/rt_amd.h declares
PROTOTYPES = {
    "rt_abi_version": (c_i, []),
    "rt_last_error": (ctypes.c_char_p, []),
    "rt_device_count": (c_i, []),
    "rt_context_create": (c_i, [c_i, P(vp)]),
    "rt_context_destroy": (None, [vp]),
    "rt_context_stream": (vp, [vp]),
    "rt_context_synchronize": (c_i, [vp]),
    "rt_default_params": (None, [P(RenderParams)]),
    "rt_camera_look_at": (None, [P(c_d), P(c_d), P(Camera)]),
    "rt_scene_create": (c_i, [P(vp)]),
    "rt_scene_destroy": (None, [vp]),
    "rt_scene_size": (c_i, [vp]),
    "rt_scene_add_triangle": (c_i, [vp, P(c_d), P(c_d), P(c_d), P(c_d), c_f]),
    "rt_scene_set_normals": (c_i, [vp, c_i, P(c_d), P(c_d), P(c_d)]),
    "rt_scene_set_texture": (c_i, [vp, c_i, P(c_d), P(c_d), P(c_d), c_i, c_i]),
    "rt_scene_add_sphere": (c_i, [vp, P(c_d), c_d, P(c_d), c_f]),
    "rt_scene_add_plane": (c_i, [vp, P(c_d), c_d, P(c_d)]),
    "rt_scene_add_cornell_box": (c_i, [vp, P(c_d), P(c_d)]),
    "rt_scene_add_cube": (c_i, [vp, P(c_d), P(c_d), P(c_d)]),
    "rt_scene_load_obj": (c_i, [vp, ctypes.c_char_p, P(c_d), c_f, c_f, c_f, c_f, c_i, c_i]),
    "rt_scene_add_texture_bmp": (c_i, [vp, ctypes.c_char_p]),
    "rt_scene_add_texture_rgb": (c_i, [vp, c_i, c_i, P(c_d)]),
    "rt_scene_build_bvh": (c_i, [vp]),
    "rt_scene_bvh_dump": (c_i, [vp, P(c_i32), P(c_i64), P(c_d), P(c_i64)]),
    "rt_scene_bvh_depth": (c_i, [vp]),
    "rt_scene_get_triangle": (c_i, [vp, c_i, P(c_d), P(c_d), P(c_d), P(c_i)]),
    "rt_scene_emit_photons": (c_i, [vp, c_i64, P(c_d), ctypes.c_uint32, c_i, c_d]),
    "rt_scene_emit_photons_gpu": (c_i, [vp, vp, c_i64, P(c_d), ctypes.c_uint32, c_i, c_d, P(Stats)]),
    "rt_scene_photon_count": (c_i64, [vp, c_i]),
    "rt_scene_photons_get": (c_i, [vp, c_i, P(c_d)]),
    "rt_scene_photons_clear": (None, [vp]),
    "rt_scene_upload": (c_i, [vp, vp]),
    "rt_scene_upload_lbvh": (c_i, [vp, vp, P(c_d)]),
    "rt_scene_export_flat": (c_i, [vp, vp, P(ctypes.c_size_t)]),
    "rt_scene_import_flat": (c_i, [vp, vp, ctypes.c_size_t]),
    "rt_scene_upload_flat": (c_i, [vp, vp, ctypes.c_size_t]),
    "rt_trace_closest": (c_i, [vp, c_i, c_i64, P(c_d), P(c_d), P(c_d), P(c_i32)]),
    "rt_trace_closest_dev": (c_i, [vp, c_i, c_i64, vp, vp, vp, vp]),
    "rt_trace_primary_dev": (c_i, [vp, P(Camera), P(RenderParams), c_i, c_i, vp, vp, vp, P(Stats)]),
    "rt_trace_primary": (c_i, [vp, P(Camera), P(RenderParams), c_i, c_i, P(c_i32), P(c_d), P(c_i32), P(Stats)]),
    "rt_photons_upload": (c_i, [vp, c_i, c_i64, P(c_d)]),
    "rt_photons_upload_dev_build": (c_i, [vp, c_i, c_i64, P(c_d)]),
    "rt_emit_photon_maps_dev": (c_i, [vp, c_i64, P(c_d), ctypes.c_uint32, c_i, c_d, P(c_i64), P(Stats)]),
    "rt_photon_map_size": (c_i64, [vp, c_i]),
    "rt_photon_map_get": (c_i, [vp, c_i, P(c_d)]),
    "rt_knn_radius": (c_i, [vp, c_i, c_i, c_d, c_i64, P(c_d), P(c_i32), P(c_d)]),
    "rt_knn_query": (c_i, [vp, c_i, c_i, c_d, c_i64, P(c_d), P(c_i32), P(c_i32)]),
    "rt_knn_estimate": (c_i, [vp, P(RenderParams), c_i64, P(c_d), P(c_d), P(c_d), P(c_d), P(c_i32)]),
    "rt_knn_estimate_grouped": (c_i, [vp, P(RenderParams), c_i64, P(c_d), P(c_d), P(c_d), P(c_d)]),
    "rt_knn_estimate_dev_noalloc": (c_i, [vp, P(RenderParams), c_i64, vp, vp, vp, vp]),
    "rt_render_rows_dev": (c_i, [vp, P(Camera), P(RenderParams), c_i, c_i, vp, vp, P(Stats)]),
    "rt_render_rows": (c_i, [vp, P(Camera), P(RenderParams), c_i, c_i, P(c_d), P(c_u8), P(Stats)]),
    "rt_gather_rows": (c_i, [P(vp), P(vp), c_i, c_i, c_i, ctypes.c_size_t, vp]),
    "rt_comm_unique_id": (c_i, [P(c_u8)]),
    "rt_comm_create": (c_i, [vp, P(c_u8), c_i, c_i, P(vp)]),
    "rt_comm_destroy": (None, [vp]),
    "rt_gather_frame": (c_i, [vp, vp, c_i, c_i, ctypes.c_size_t, vp]),
    "rt_save_bmp": (c_i, [ctypes.c_char_p, c_i, c_i, c_i, P(c_d)]),
    "rt_save_bmp_bgr8": (c_i, [ctypes.c_char_p, c_i, c_i, c_i, P(c_u8)]),
}


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` (or __graft_entry__.build())")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()
