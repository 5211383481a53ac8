"""ctypes binding of librtmi.so (include/rtmi.h).  The library is the product: there is no fallback, and a missing
or unloadable library raises."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTMI_LIB") or os.path.join(_HERE, "librtmi.so")  # RTMI_LIB: an alternative build of the same library

ACCUM_PLANES = 22
COMM_ID_BYTES = 128
RESOLVE_PLANES = 15
INVALID_ID = 0xFFFFFFFF

RTMI_OK = 0
ERRORS = {-1: "RTMI_ERR_ARG", -2: "RTMI_ERR_DEVICE", -3: "RTMI_ERR_STATE", -4: "RTMI_ERR_IO",
          -5: "RTMI_ERR_CANCELLED", -6: "RTMI_ERR_LIMIT"}


class RtmiError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s: %s" % (ERRORS.get(code, code), message))
        self.code = code


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("rgb", C.c_float * 3), ("ior", C.c_float), ("roughness", C.c_float),
                ("pad", C.c_uint32 * 2)]


class Camera(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("half_width", C.c_float),
                ("half_height", C.c_float), ("focal_length", C.c_float), ("aperture", C.c_float),
                ("lens", C.c_float * 3), ("rot", C.c_float * 4)]


class TransformPOD(C.Structure):
    _fields_ = [("translation", C.c_float * 3), ("scale", C.c_float * 3), ("rot", C.c_float * 4)]


class RenderParams(C.Structure):
    _fields_ = [("spp", C.c_uint32), ("sample_begin", C.c_uint32), ("sample_end", C.c_uint32),
                ("max_depth", C.c_uint32), ("integrator", C.c_uint32), ("samples_per_pass", C.c_uint32),
                ("seed", C.c_uint64), ("allowed_error", C.c_float), ("flags", C.c_uint32)]


class RenderStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64), ("passes", C.c_uint32), ("max_bounces", C.c_uint32),
                ("gpu_ms", C.c_float), ("bounce_ms", C.c_float), ("bounce_launches", C.c_uint32), ("extend_ms", C.c_float),
                ("tail_rays", C.c_uint64)]


class BounceProfile(C.Structure):
    _fields_ = [("depth", C.c_uint32), ("kernel", C.c_uint32), ("rays", C.c_uint64), ("extend_ms", C.c_float),
                ("shade_ms", C.c_float), ("launches", C.c_uint32), ("reserved", C.c_uint32)]


class SceneInfo(C.Structure):
    _fields_ = [("n_prims", C.c_uint64), ("n_nodes", C.c_uint64), ("node_bytes", C.c_uint64),
                ("prim_bytes", C.c_uint64), ("n_geoms", C.c_uint32), ("n_materials", C.c_uint32),
                ("stack_need", C.c_uint32), ("max_depth", C.c_uint32), ("build_ms", C.c_float),
                ("sah_cost", C.c_float), ("n_records", C.c_uint32), ("spatial_splits", C.c_uint32)]


RAY_DTYPE = np.dtype([("org", np.float32, 3), ("tnear", np.float32), ("dir", np.float32, 3), ("tfar", np.float32)])
HIT_DTYPE = np.dtype([("t", np.float32), ("pos", np.float32, 3), ("n", np.float32, 3), ("uv", np.float32, 2),
                      ("geom_id", np.uint32), ("prim_id", np.uint32), ("mat_id", np.uint32)])

PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_double, C.c_void_p)

# every symbol include/rtmi.h declares: name -> (restype, argtypes)
_vp, _u32, _sz, _f32 = C.c_void_p, C.c_uint32, C.c_size_t, C.c_float
_fp, _u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
SIGNATURES = {
    "rtmi_last_error": (C.c_char_p, []),
    "rtmi_version": (C.c_char_p, []),
    "rtmi_device_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "rtmi_device_destroy": (None, [_vp]),
    "rtmi_device_stream": (_vp, [_vp]),
    "rtmi_device_count_visits": (C.c_int, [_vp, C.c_int]),
    "rtmi_device_get_visits": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "rtmi_device_last_render_profile": (C.c_int, [_vp, _vp, _u32, _u32p]),
    "rtmi_scene_create": (C.c_int, [_vp, C.POINTER(_vp)]),
    "rtmi_scene_destroy": (None, [_vp]),
    "rtmi_scene_insert_material": (C.c_int, [_vp, C.POINTER(Material), _u32p]),
    "rtmi_scene_insert_light": (C.c_int, [_vp, _fp]),
    "rtmi_scene_insert_mesh": (C.c_int, [_vp, _u32, _vp, _sz, _vp, _sz, _u32p]),
    "rtmi_scene_insert_sphere": (C.c_int, [_vp, _u32, _fp, _f32, _u32p]),
    "rtmi_scene_load_obj": (C.c_int, [_vp, C.c_char_p, C.POINTER(TransformPOD), _u32, _u32p, _u32p]),
    "rtmi_scene_commit": (C.c_int, [_vp, PROGRESS_FN, _vp]),
    "rtmi_scene_bounds": (C.c_int, [_vp, _fp, _fp]),
    "rtmi_scene_get_info": (C.c_int, [_vp, C.POINTER(SceneInfo)]),
    "rtmi_scene_geometry_material": (C.c_int, [_vp, _u32, _u32p]),
    "rtmi_intersect": (C.c_int, [_vp, _vp, _vp, _sz]),
    "rtmi_intersect_device": (C.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "rtmi_intersect_device_static": (C.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "rtmi_film_create": (C.c_int, [_vp, _vp, _sz, _vp, C.POINTER(_vp)]),
    "rtmi_film_create_tiles": (C.c_int, [_vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32, C.POINTER(_vp)]),
    "rtmi_film_create_frame": (C.c_int, [_vp, _u32, _u32, _u32, _u32, C.POINTER(_vp)]),
    "rtmi_film_pixel_count": (_sz, [_vp]),
    "rtmi_film_pixels": (C.c_int, [_vp, _vp]),
    "rtmi_film_gather": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "rtmi_comm_gather_abort": (C.c_int, [_vp]),
    "rtmi_comm_unique_id": (C.c_int, [C.c_char_p]),
    "rtmi_comm_create": (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int, C.POINTER(_vp)]),
    "rtmi_comm_destroy": (None, [_vp]),
    "rtmi_tile_partition": (C.c_int, [_u32, _u32, _u32, _u32p, _u32p, _u32p]),
    "rtmi_film_destroy": (None, [_vp]),
    "rtmi_film_clear": (C.c_int, [_vp]),
    "rtmi_film_accum_device": (_vp, [_vp]),
    "rtmi_film_download": (C.c_int, [_vp, _vp]),
    "rtmi_film_resolve": (C.c_int, [_vp, _vp, C.c_int]),
    "rtmi_render": (C.c_int, [_vp, _vp, C.POINTER(Camera), C.POINTER(RenderParams), C.POINTER(RenderStats)]),
    "rtmi_tile_count": (_u32, [_u32, _u32, _u32]),
    "rtmi_tile_get": (C.c_int, [_u32, _u32, _u32, _u32, _u32, _u32, _u32p]),
}

_lib = None


def load():
    """Loads librtmi.so.  Raises when the library is absent: the HIP path is the only path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "%s not found: build it with `make -C raytracing_amd/csrc` (or __graft_entry__.build())" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != RTMI_OK:
        raise RtmiError(rc, load().rtmi_last_error().decode("utf-8", "replace"))
