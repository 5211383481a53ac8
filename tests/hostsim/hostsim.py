"""ctypes front end of the host debugging build of the device code (tests/hostsim/hostsim.cpp).  Test harness only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libhostsim.so")

RAY_DTYPE = np.dtype([("org", np.float32, 3), ("tnear", np.float32), ("dir", np.float32, 3), ("tfar", np.float32)])
HIT_DTYPE = np.dtype([("t", np.float32), ("pos", np.float32, 3), ("n", np.float32, 3), ("uv", np.float32, 2),
                      ("geom_id", np.uint32), ("prim_id", np.uint32), ("mat_id", np.uint32)])
MAT_KINDS = {"DiffuseBxDF": 0, "DielectricBxDF": 1, "ThinDielectricBxDF": 2, "EmitBxDF": 3}


class CameraView(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("half_width", C.c_float),
                ("half_height", C.c_float), ("focal_length", C.c_float), ("aperture", C.c_float),
                ("lens", C.c_float * 3), ("rot", C.c_float * 4)]


class Params(C.Structure):
    _fields_ = [("spp", C.c_uint32), ("sample_begin", C.c_uint32), ("sample_end", C.c_uint32),
                ("max_depth", C.c_uint32), ("integrator", C.c_uint32), ("samples_per_pass", C.c_uint32),
                ("seed", C.c_uint64), ("allowed_error", C.c_float), ("flags", C.c_uint32)]


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True, capture_output=True)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        vp, u32, sz, f32 = C.c_void_p, C.c_uint32, C.c_size_t, C.c_float
        fp = C.POINTER(C.c_float)
        L.hs_scene_new.restype = vp
        L.hs_scene_free.argtypes = [vp]
        L.hs_insert_material.restype = u32
        L.hs_insert_material.argtypes = [vp, u32, fp, f32, f32]
        L.hs_insert_mesh.restype = u32
        L.hs_insert_mesh.argtypes = [vp, u32, vp, sz, vp, sz]
        L.hs_insert_sphere.restype = u32
        L.hs_insert_sphere.argtypes = [vp, u32, fp, f32]
        L.hs_load_obj.restype = C.c_int
        L.hs_load_obj.argtypes = [vp, C.c_char_p, fp, fp, fp, u32]
        L.hs_commit.restype = C.c_int
        L.hs_commit.argtypes = [vp, C.POINTER(u32)]
        L.hs_bounds.argtypes = [vp, fp, fp]
        L.hs_tree_stats.argtypes = [vp, C.POINTER(u32)]
        L.hs_trace.restype = C.c_size_t
        L.hs_trace.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp]
        L.hs_stack_high.restype = C.c_int
        L.hs_stack_high.argtypes = [vp]
        L.hs_items.restype = C.c_size_t
        L.hs_items.argtypes = [vp, vp, C.c_size_t]
        L.hs_geometry_triangles.restype = sz
        L.hs_geometry_triangles.argtypes = [vp, u32, vp, sz]
        L.hs_geometry_material.restype = u32
        L.hs_geometry_material.argtypes = [vp, u32]
        L.hs_geometry_count.restype = u32
        L.hs_geometry_count.argtypes = [vp]
        L.hs_material_count.restype = u32
        L.hs_material_count.argtypes = [vp]
        L.hs_material_get.argtypes = [vp, u32, C.POINTER(u32), fp]
        L.hs_intersect.argtypes = [vp, vp, vp, sz]
        L.hs_render_sample.restype = u32
        L.hs_render_sample.argtypes = [vp, C.POINTER(CameraView), C.POINTER(Params), u32, u32, u32, fp]
        L.hs_render.restype = C.c_uint64
        L.hs_render.argtypes = [vp, C.POINTER(CameraView), C.POINTER(Params), vp, sz, vp]
        _lib = L
    return _lib


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def camera_view(cam):
    c = CameraView()
    c.width, c.height = int(cam.width), int(cam.height)
    c.half_width = float(cam.viewport_half_width)
    c.half_height = float(cam.viewport_half_height)
    c.focal_length = float(cam.focal_length)
    c.aperture = float(cam.aperture)
    c.lens = _f3(cam.center_of_lens)
    c.rot = (C.c_float * 4)(*[float(x) for x in cam.rotation])
    return c


def params(spp, sample_begin=0, sample_end=None, max_depth=64, integrator=0, seed=0, flags=0):
    p = Params()
    p.spp = spp
    p.sample_begin = sample_begin
    p.sample_end = spp if sample_end is None else sample_end
    p.max_depth = max_depth
    p.integrator = integrator
    p.seed = seed
    p.flags = flags
    return p


class HostSimScene:
    """SceneT surface over the host build of the product's scene assembly + device code."""

    def __init__(self):
        self._l = lib()
        self._h = self._l.hs_scene_new()
        self.materials = ["Sky"]
        self.lights = []
        self.info = None

    def __del__(self):
        if getattr(self, "_h", None):
            self._l.hs_scene_free(self._h)
            self._h = None

    def insert_material(self, desc):
        bxdf = getattr(desc, "material", desc)
        rgb = getattr(bxdf, "albedo", None)
        if rgb is None:
            rgb = getattr(bxdf, "le", (0.0, 0.0, 0.0))
        self.materials.append(desc)
        return int(self._l.hs_insert_material(self._h, MAT_KINDS[type(bxdf).__name__], _f3(rgb),
                                              float(getattr(bxdf, "ior", 0.0)),
                                              float(getattr(bxdf, "roughness", 0.0))))

    def insert_light(self, light):
        self.lights.append(tuple(getattr(light, "light_pos", light)))

    def insert_mesh(self, material, vertices, indices):
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        return int(self._l.hs_insert_mesh(self._h, int(material), v.ctypes.data, v.shape[0], i.ctypes.data,
                                          i.shape[0]))

    def insert_sphere(self, material, origin, radius):
        return int(self._l.hs_insert_sphere(self._h, int(material), _f3(origin), float(radius)))

    def load_obj(self, path, transform, default_material):
        n = self._l.hs_load_obj(self._h, str(path).encode(), _f3(transform.translation), _f3(transform.scale),
                                (C.c_float * 4)(*[float(x) for x in transform.rot]), int(default_material))
        if n < 0:
            raise IOError("hs_load_obj failed for %s" % path)
        return n

    def commit(self):
        out = (C.c_uint32 * 4)()
        if self._l.hs_commit(self._h, out):
            raise RuntimeError("hs_commit failed")
        stats = (C.c_uint32 * 2)()
        self._l.hs_tree_stats(self._h, stats)
        self.info = {"n_prims": out[0], "n_nodes": out[1], "stack_need": out[2], "max_depth": out[3],
                     "n_records": stats[0], "spatial_splits": stats[1]}
        return self

    def bounding_box(self):
        lo, hi = (C.c_float * 3)(), (C.c_float * 3)()
        self._l.hs_bounds(self._h, lo, hi)
        return np.array(lo[:], dtype=np.float32), np.array(hi[:], dtype=np.float32)

    def stack_high(self):
        return int(self._l.hs_stack_high(self._h))

    def items(self):
        """The item array of the committed tree (nodes and primitive records), one row of 16 words per 64-byte item."""
        n = int(self._l.hs_items(self._h, None, 0))
        out = np.zeros((n, 16), dtype=np.uint32)
        self._l.hs_items(self._h, out.ctypes.data, n)
        return out

    def visits(self, rays):
        """(inner-node visits, record tests) per ray of the closest-hit traversal of `rays`."""
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        n = rays.shape[0]
        offs = np.zeros(n + 1, dtype=np.uint64)
        total = int(self._l.hs_trace(self._h, rays.ctypes.data, n, None, 0, offs.ctypes.data))
        buf = np.zeros(max(total, 1), dtype=np.uint32)
        self._l.hs_trace(self._h, rays.ctypes.data, n, buf.ctypes.data, total, offs.ctypes.data)
        rec = int(((buf[:total] & 0x80000000) != 0).sum())
        return (total - rec) / n, rec / n

    def geometry_count(self):
        return int(self._l.hs_geometry_count(self._h))

    def geometry_material(self, g):
        return int(self._l.hs_geometry_material(self._h, g))

    def geometry_triangles(self, g):
        n = self._l.hs_geometry_triangles(self._h, g, None, 0)
        out = np.zeros((n, 9), dtype=np.float32)
        if n:
            self._l.hs_geometry_triangles(self._h, g, out.ctypes.data, n)
        return out

    def material_table(self):
        out = []
        for i in range(self._l.hs_material_count(self._h)):
            k = C.c_uint32()
            v = (C.c_float * 5)()
            self._l.hs_material_get(self._h, i, C.byref(k), v)
            out.append((int(k.value),) + tuple(v[:]))
        return out

    def intersect(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        self._l.hs_intersect(self._h, rays.ctypes.data, hits.ctypes.data, rays.shape[0])
        return hits

    def render_sample(self, camera, prm, x, y, sample):
        out = (C.c_float * 15)()
        cam = camera_view(camera)
        rays = self._l.hs_render_sample(self._h, C.byref(cam), C.byref(prm), x, y, sample, out)
        return np.array(out[:], dtype=np.float32), int(rays)

    def render(self, camera, prm, xy, accum=None):
        xy = np.ascontiguousarray(xy, dtype=np.uint32).reshape(-1, 2)
        n = xy.shape[0]
        if accum is None:
            accum = np.zeros((22, n), dtype=np.float32)
        cam = camera_view(camera)
        rays = self._l.hs_render(self._h, C.byref(cam), C.byref(prm), xy.ctypes.data, n, accum.ctypes.data)
        return accum, int(rays)
