"""ctypes front end of the CPU oracle (oracle/rt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
the product package.  The scene class offers the same build/query surface as the reference's SceneT / Shape
(crates/rt/src/scene/mod.rs:8-26, crates/rt/src/shape/mod.rs:3-9), so scene descriptions written against that
surface can be fed to the oracle and to the HIP path alike.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import obj_tobj

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "librt_oracle.so")

ACCUM_PLANES = 22
RESOLVE_PLANES = 15
INVALID_ID = 0xFFFFFFFF
ISECT_BRUTE, ISECT_BVH = 0, 1
FORM_ITERATIVE, FORM_RECURSIVE = 0, 1
INTEGRATOR_PATH_TRACER, INTEGRATOR_RANDOM_WALK = 0, 1
MAT_KINDS = {"DiffuseBxDF": 0, "DielectricBxDF": 1, "ThinDielectricBxDF": 2, "EmitBxDF": 3}


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("rgb", C.c_float * 3), ("ior", C.c_float), ("roughness", C.c_float),
                ("pad", C.c_uint32 * 2)]


class Camera(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("half_width", C.c_float),
                ("half_height", C.c_float), ("focal_length", C.c_float), ("aperture", C.c_float),
                ("lens", C.c_float * 3), ("rot", C.c_float * 4)]


class RenderParams(C.Structure):
    _fields_ = [("spp", C.c_uint32), ("sample_begin", C.c_uint32), ("sample_end", C.c_uint32),
                ("max_depth", C.c_uint32), ("integrator", C.c_uint32), ("isect_mode", C.c_uint32),
                ("form", C.c_uint32), ("threads", C.c_uint32), ("seed", C.c_uint64), ("allowed_error", C.c_float),
                ("flags", C.c_uint32)]


class Tile(C.Structure):
    _fields_ = [("x_start", C.c_uint32), ("x_end", C.c_uint32), ("y_start", C.c_uint32), ("y_end", C.c_uint32)]


RAY_DTYPE = np.dtype([("org", np.float32, 3), ("tnear", np.float32), ("dir", np.float32, 3), ("tfar", np.float32)])
HIT_DTYPE = np.dtype([("t", np.float32), ("pos", np.float32, 3), ("n", np.float32, 3), ("uv", np.float32, 2),
                      ("geom_id", np.uint32), ("prim_id", np.uint32), ("mat_id", np.uint32)])


def build(force=False):
    """Compiles oracle/rt_oracle.c with gcc (the Makefile beside it)."""
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("rt_oracle.c", "rt_oracle.h")):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    vp, u32, u64, f32, sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_size_t
    fp = C.POINTER(C.c_float)
    sig = {
        "orc_scene_new": (vp, []),
        "orc_scene_free": (None, [vp]),
        "orc_scene_insert_material": (u32, [vp, C.POINTER(Material)]),
        "orc_scene_insert_light": (None, [vp, fp]),
        "orc_scene_insert_mesh": (u32, [vp, u32, vp, sz, vp, sz]),
        "orc_scene_insert_sphere": (u32, [vp, u32, fp, f32]),
        "orc_scene_commit": (C.c_int, [vp]),
        "orc_scene_bounds": (None, [vp, fp, fp]),
        "orc_scene_prim_count": (sz, [vp]),
        "orc_intersect": (None, [vp, vp, vp, sz, C.c_int]),
        "orc_render": (u64, [vp, C.POINTER(Camera), C.POINTER(RenderParams), vp, sz, vp]),
        "orc_render_sample": (u32, [vp, C.POINTER(Camera), C.POINTER(RenderParams), u32, u32, u32, fp]),
        "orc_resolve": (None, [vp, sz, vp, C.c_int]),
        "orc_tile_count": (u32, [u32, u32, u32]),
        "orc_tile_get": (C.c_int, [u32, u32, u32, u32, u32, u32, C.POINTER(Tile)]),
        "orc_philox4x32_10": (None, [C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]),
        "orc_uniform4": (None, [u64, u32, u32, u32, u32, fp]),
        "orc_sincos_2pi": (None, [f32, fp, fp]),
        "orc_frame": (None, [fp, fp]),
        "orc_bsdf_sample": (C.c_int, [C.POINTER(Material), fp, fp, fp, fp, fp, fp]),
        "orc_bsdf_f": (None, [C.POINTER(Material), fp, fp, fp, fp]),
        "orc_fresnel_dielectric": (f32, [f32, f32]),
        "orc_refract": (C.c_int, [fp, fp, f32, fp, fp]),
        "orc_camera_ray": (None, [C.POINTER(Camera), C.POINTER(RenderParams), u32, u32, u32, vp]),
        "orc_ray_triangle": (C.c_int, [vp, fp, fp, fp, fp, fp, fp]),
        "orc_ray_sphere": (C.c_int, [vp, fp, f32, fp, fp]),
        "orc_slab_test": (C.c_int, [fp, fp, vp, fp, fp]),
        "orc_srgb_from_linear": (f32, [f32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def make_material(desc):
    """Accepts a MaterialDescriptor-like object (has .material) or a BxDF-like object, by class name."""
    bxdf = getattr(desc, "material", desc)
    kind = MAT_KINDS[type(bxdf).__name__]
    m = Material()
    m.kind = kind
    rgb = getattr(bxdf, "albedo", None)
    if rgb is None:
        rgb = getattr(bxdf, "le", (0.0, 0.0, 0.0))
    m.rgb = (C.c_float * 3)(*[float(x) for x in rgb])
    m.ior = float(getattr(bxdf, "ior", 0.0))
    m.roughness = float(getattr(bxdf, "roughness", 0.0))
    return m


def make_camera(cam):
    c = Camera()
    c.width, c.height = int(cam.width), int(cam.height)
    c.half_width = float(cam.viewport_half_width)
    c.half_height = float(cam.viewport_half_height)
    c.focal_length = float(cam.focal_length)
    c.aperture = float(cam.aperture)
    c.lens = f3(cam.center_of_lens)
    c.rot = (C.c_float * 4)(*[float(x) for x in cam.rotation])
    return c


SAMPLER_STRATIFIED, SAMPLER_UNIFORM, SAMPLER_PIXEL_CENTRE = 0x00, 0x01, 0x02
FILTER_BOX, FILTER_TRIANGLE, FILTER_NONE = 0x00, 0x10, 0x20


def make_params(spp, sample_begin=0, sample_end=None, max_depth=64, integrator=0, isect_mode=ISECT_BVH,
                form=FORM_ITERATIVE, threads=0, seed=0, allowed_error=0.0, flags=0):
    p = RenderParams()
    p.spp = spp
    p.sample_begin = sample_begin
    p.sample_end = spp if sample_end is None else sample_end
    p.max_depth = max_depth
    p.integrator = integrator
    p.isect_mode = isect_mode
    p.form = form
    p.threads = threads
    p.seed = seed
    p.allowed_error = allowed_error or 0.0
    p.flags = flags
    return p


class _DiffuseBxDF:
    def __init__(self, albedo):
        self.albedo = albedo


_DiffuseBxDF.__name__ = "DiffuseBxDF"


class OracleScene:
    """SceneT + Shape over the oracle (crates/rt/src/aggregate/embree.rs:19-266 restated on the CPU)."""

    def __init__(self):
        self._l = lib()
        self._h = self._l.orc_scene_new()
        self.materials = ["Sky"]
        self.lights = []

    def __del__(self):
        if getattr(self, "_h", None):
            self._l.orc_scene_free(self._h)
            self._h = None

    # -- SceneT
    def insert_material(self, desc):
        m = make_material(desc)
        self.materials.append(desc)
        return int(self._l.orc_scene_insert_material(self._h, C.byref(m)))

    def insert_light(self, light):
        pos = getattr(light, "light_pos", light)
        self.lights.append(tuple(pos))
        self._l.orc_scene_insert_light(self._h, f3(pos))

    def insert_mesh(self, material, vertices, indices):
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        return int(self._l.orc_scene_insert_mesh(self._h, int(material), v.ctypes.data, v.shape[0], i.ctypes.data,
                                                 i.shape[0]))

    def insert_sphere(self, material, origin, radius):
        return int(self._l.orc_scene_insert_sphere(self._h, int(material), f3(origin), float(radius)))

    # -- ObjLoaderExt (crates/rt/src/loader/obj.rs:24-123)
    def load_obj(self, path, transform, default_material):
        models, materials = obj_tobj.load_obj(path)
        ids = []
        if materials is not None:
            for m in materials:
                ids.append(self.insert_material(_DiffuseBxDF(m.diffuse)))
        handles = []
        for model in models:
            if materials is not None and model.material_id is not None and model.material_id < len(ids):
                mat = ids[model.material_id]
            else:
                mat = default_material
            pos = obj_tobj.apply_transform(model.positions, transform.translation, transform.scale, transform.rot)
            handles.append(self.insert_mesh(mat, pos, model.indices))
        return handles

    def commit(self, progress=None):
        self._l.orc_scene_commit(self._h)
        if progress is not None:
            progress(1.0)
        return self

    def into_world(self):
        return self

    # -- Shape
    def bounding_box(self):
        lo = (C.c_float * 3)()
        hi = (C.c_float * 3)()
        self._l.orc_scene_bounds(self._h, lo, hi)
        return np.array(lo[:], dtype=np.float32), np.array(hi[:], dtype=np.float32)

    def prim_count(self):
        return int(self._l.orc_scene_prim_count(self._h))

    def intersect(self, rays, mode=ISECT_BVH):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        self._l.orc_intersect(self._h, rays.ctypes.data, hits.ctypes.data, rays.shape[0], mode)
        return hits

    # -- Executor
    def render(self, camera, params, xy, accum=None):
        xy = np.ascontiguousarray(xy, dtype=np.uint32).reshape(-1, 2)
        n = xy.shape[0]
        if accum is None:
            accum = np.zeros((ACCUM_PLANES, n), dtype=np.float32)
        cam = camera if isinstance(camera, Camera) else make_camera(camera)
        rays = self._l.orc_render(self._h, C.byref(cam), C.byref(params), xy.ctypes.data, n, accum.ctypes.data)
        return accum, int(rays)

    def render_sample(self, camera, params, x, y, sample):
        cam = camera if isinstance(camera, Camera) else make_camera(camera)
        out = (C.c_float * 15)()
        rays = self._l.orc_render_sample(self._h, C.byref(cam), C.byref(params), x, y, sample, out)
        return np.array(out[:], dtype=np.float32), int(rays)

    def camera_rays(self, camera, params, xy, sample):
        cam = camera if isinstance(camera, Camera) else make_camera(camera)
        xy = np.asarray(xy, dtype=np.uint32).reshape(-1, 2)
        rays = np.zeros(xy.shape[0], dtype=RAY_DTYPE)
        one = np.zeros(1, dtype=RAY_DTYPE)
        for i, (x, y) in enumerate(xy):
            self._l.orc_camera_ray(C.byref(cam), C.byref(params), int(x), int(y), int(sample), one.ctypes.data)
            rays[i] = one[0]
        return rays


def resolve(accum, srgb=False):
    accum = np.ascontiguousarray(accum, dtype=np.float32)
    n = accum.shape[1]
    out = np.zeros((RESOLVE_PLANES, n), dtype=np.float32)
    lib().orc_resolve(accum.ctypes.data, n, out.ctypes.data, 1 if srgb else 0)
    return out


def tiles(width, height, grain, offset_x=0, offset_y=0):
    L = lib()
    out = []
    t = Tile()
    for i in range(L.orc_tile_count(width, height, grain)):
        L.orc_tile_get(offset_x, offset_y, width, height, grain, i, C.byref(t))
        out.append((t.x_start, t.x_end, t.y_start, t.y_end))
    return out


def philox(ctr, key):
    u32 = C.c_uint32
    out = (u32 * 4)()
    lib().orc_philox4x32_10((u32 * 4)(*ctr), (u32 * 2)(*key), out)
    return tuple(out[:])


def uniform4(seed, x, y, sample, stream):
    out = (C.c_float * 4)()
    lib().orc_uniform4(seed, x, y, sample, stream, out)
    return np.array(out[:], dtype=np.float32)


def sincos_2pi(u):
    s = C.c_float()
    c = C.c_float()
    lib().orc_sincos_2pi(float(u), C.byref(s), C.byref(c))
    return s.value, c.value
