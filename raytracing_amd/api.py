"""Host-side mirror of the reference's scene / shape / executor interface over the C ABI (include/rtmi.h).

  HipScene            SceneT + ObjLoaderExt + commit         crates/rt/src/scene/mod.rs:8-26, aggregate/embree.rs:19-79,
                                                             loader/obj.rs:15-22
  CommittedHipScene   Shape (+ into_world)                   crates/rt/src/shape/mod.rs:3-9, aggregate/embree.rs:81-162
  Tile / Tiler        tile rectangles                        crates/rt-cli/src/tile.rs:4-97
  TileMsg             tile + per-pixel channel values        crates/rt-cli/src/executor.rs:41-44
  Executor            run_multithreaded / run_monothreaded   crates/rt-cli/src/executor.rs:46-205
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np

from . import _lib
from ._lib import ACCUM_PLANES, COMM_ID_BYTES, HIT_DTYPE, RAY_DTYPE, RESOLVE_PLANES, RtmiError, check
from .types import Camera, RenderRange

MAT_KINDS = {"DiffuseBxDF": 0, "DielectricBxDF": 1, "ThinDielectricBxDF": 2, "EmitBxDF": 3}
INTEGRATORS = {"path-tracer": 0, "basic": 1}

# channel order of PixelRenderResult (crates/rt/src/renderer.rs:51-59) and its planes in a resolved image
CHANNELS = {"Normal": slice(0, 3), "Position": slice(3, 6), "Albedo": slice(6, 9), "Color": slice(9, 12),
            "Variance": slice(12, 13), "Z": slice(13, 14), "RayDepth": slice(14, 15)}


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class Device:
    """embree4_rs::Device stand-in (crates/rt-cli/src/main.rs:74-90): one HIP device + the stream the library uses."""

    def __init__(self, ordinal: int = -1):
        self._l = _lib.load()
        h = C.c_void_p()
        check(self._l.rtmi_device_create(int(ordinal), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._l.rtmi_device_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return self._l.rtmi_device_stream(self._h)

    def count_visits(self, enable=True):
        check(self._l.rtmi_device_count_visits(self._h, 1 if enable else 0))

    KERNELS = ("extend_kernel", "extend_queue_kernel", "tail_kernel")

    def last_render_profile(self):
        """rtmi_device_last_render_profile: per bounce, the kernel that traced it, its rays and the device time inside
        its closest-hit and shading launches, for the last render on this device."""
        n = C.c_uint32()
        check(self._l.rtmi_device_last_render_profile(self._h, None, 0, C.byref(n)))
        buf = (_lib.BounceProfile * max(n.value, 1))()
        check(self._l.rtmi_device_last_render_profile(self._h, buf, n.value, C.byref(n)))
        return [{"depth": int(b.depth), "kernel": self.KERNELS[b.kernel], "rays": int(b.rays),
                 "extend_ms": float(b.extend_ms), "shade_ms": float(b.shade_ms), "launches": int(b.launches)}
                for b in buf[:n.value]]

    def visits(self):
        out = (C.c_uint64 * 2)()
        check(self._l.rtmi_device_get_visits(self._h, out))
        return int(out[0]), int(out[1])


def _material_pod(desc):
    bxdf = getattr(desc, "material", desc)
    m = _lib.Material()
    m.kind = MAT_KINDS[type(bxdf).__name__]
    rgb = getattr(bxdf, "albedo", None)
    if rgb is None:
        rgb = getattr(bxdf, "le", (0.0, 0.0, 0.0))
    m.rgb = _f3(rgb)
    m.ior = float(getattr(bxdf, "ior", 0.0))
    m.roughness = float(getattr(bxdf, "roughness", 0.0))
    return m


def camera_pod(cam: Camera):
    c = _lib.Camera()
    c.width, c.height = int(cam.width), int(cam.height)
    c.half_width = float(cam.viewport_half_width)
    c.half_height = float(cam.viewport_half_height)
    c.focal_length = float(cam.focal_length)
    c.aperture = float(cam.aperture)
    c.lens = _f3(cam.center_of_lens)
    c.rot = (C.c_float * 4)(*[float(x) for x in cam.rotation])
    return c


class HipScene:
    """EmbreeScene's replacement: collects materials, lights, meshes and spheres, then commit() builds the BVH and
    uploads it to the GPU."""

    def __init__(self, device: Device):
        self.device = device
        self._l = device._l
        h = C.c_void_p()
        check(self._l.rtmi_scene_create(device._h, C.byref(h)))
        self._h = h
        self.materials = ["Sky"]  # embree.rs:41-46
        self.lights = []

    def close(self):
        if getattr(self, "_h", None):
            self._l.rtmi_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- SceneT
    def insert_material(self, mat) -> int:
        out = C.c_uint32()
        pod = _material_pod(mat)
        check(self._l.rtmi_scene_insert_material(self._h, C.byref(pod), C.byref(out)))
        self.materials.append(mat)
        return int(out.value)

    def insert_light(self, light):
        pos = getattr(light, "light_pos", light)
        check(self._l.rtmi_scene_insert_light(self._h, _f3(pos)))
        self.lights.append(tuple(pos))

    def insert_mesh(self, material: int, vertices, indices) -> int:
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        out = C.c_uint32()
        check(self._l.rtmi_scene_insert_mesh(self._h, int(material), v.ctypes.data, v.shape[0], i.ctypes.data,
                                             i.shape[0], C.byref(out)))
        return int(out.value)

    def insert_sphere(self, material: int, origin, radius: float) -> int:
        out = C.c_uint32()
        check(self._l.rtmi_scene_insert_sphere(self._h, int(material), _f3(origin), float(radius), C.byref(out)))
        return int(out.value)

    # -- ObjLoaderExt
    def load_obj(self, mesh_path, transform, default_material: int):
        t = _lib.TransformPOD()
        t.translation = _f3(transform.translation)
        t.scale = _f3(transform.scale)
        t.rot = (C.c_float * 4)(*[float(x) for x in transform.rot])
        first, count = C.c_uint32(), C.c_uint32()
        check(self._l.rtmi_scene_load_obj(self._h, str(mesh_path).encode(), C.byref(t), int(default_material),
                                          C.byref(first), C.byref(count)))
        return list(range(first.value, first.value + count.value))

    # -- commit / commit_with_progress
    def commit(self) -> "CommittedHipScene":
        return self.commit_with_progress(None)

    def commit_with_progress(self, progress_callback: Optional[Callable[[float], bool]]) -> "CommittedHipScene":
        if progress_callback is None:
            cb = C.cast(None, _lib.PROGRESS_FN)
        else:
            cb = _lib.PROGRESS_FN(lambda amount, _user: 1 if progress_callback(amount) in (True, None) else 0)
        check(self._l.rtmi_scene_commit(self._h, cb, None))
        return CommittedHipScene(self)


class CommittedHipScene:
    """The committed, immutable scene: answers closest-hit queries on the GPU."""

    def __init__(self, scene: HipScene):
        self.scene = scene
        self._l = scene._l
        self._h = scene._h

    def into_world(self):
        """World { objects, lights, materials, world_material } (crates/rt/src/renderer.rs:154-159)."""
        return self

    @property
    def objects(self):
        return self

    @property
    def lights(self):
        return self.scene.lights

    @property
    def materials(self):
        return self.scene.materials

    world_material = 0

    def info(self):
        out = _lib.SceneInfo()
        check(self._l.rtmi_scene_get_info(self._h, C.byref(out)))
        return {k: getattr(out, k) for k, _ in out._fields_}

    def geometry_material(self, geom: int) -> int:
        out = C.c_uint32()
        check(self._l.rtmi_scene_geometry_material(self._h, int(geom), C.byref(out)))
        return int(out.value)

    # -- Shape
    def intersection_full(self, ray):
        """One ray (org, dir, bounds); returns the hit record or None.  Batches should use intersect()."""
        rays = np.zeros(1, dtype=RAY_DTYPE)
        rays["org"][0], rays["dir"][0] = ray[0], ray[1]
        rays["tnear"][0], rays["tfar"][0] = ray[2] if len(ray) > 2 else (0.0, math.inf)
        hit = self.intersect(rays)[0]
        return None if hit["geom_id"] == _lib.INVALID_ID else hit

    def intersect_bare(self, ray):
        raise NotImplementedError("unimplemented in the reference too (aggregate/embree.rs:142-144)")

    def bounding_box(self):
        lo, hi = (C.c_float * 3)(), (C.c_float * 3)()
        check(self._l.rtmi_scene_bounds(self._h, lo, hi))
        return np.array(lo[:], dtype=np.float32), np.array(hi[:], dtype=np.float32)

    def intersect(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        check(self._l.rtmi_intersect(self._h, rays.ctypes.data, hits.ctypes.data, rays.shape[0]))
        return hits

    def intersect_device(self, d_rays: int, d_hits: int, n: int, stream=None):
        check(self._l.rtmi_intersect_device(self._h, C.c_void_p(d_rays), C.c_void_p(d_hits), n, stream))

    def intersect_device_static(self, d_rays: int, d_hits: int, n: int, stream=None):
        """intersect_device without any device allocation, ever (rtmi_intersect_device_static)."""
        check(self._l.rtmi_intersect_device_static(self._h, C.c_void_p(d_rays), C.c_void_p(d_hits), n, stream))


@dataclass(frozen=True)
class Tile:
    x_start: int
    x_end: int
    y_start: int
    y_end: int

    def width(self):
        return self.x_end - self.x_start

    def height(self):
        return self.y_end - self.y_start

    def __len__(self):
        return self.width() * self.height()

    def pixels(self):
        """Row-major (x, y) pairs, the order of Tile's iterator (tile.rs:28-40)."""
        ys, xs = np.mgrid[self.y_start:self.y_end, self.x_start:self.x_end]
        return np.stack([xs.ravel(), ys.ravel()], 1).astype(np.uint32)


@dataclass(frozen=True)
class Tiler:
    offset_x: int
    offset_y: int
    width: int
    height: int
    x_grainsize: int
    y_grainsize: int

    def tile_dimensions(self):
        return (-(-self.width // self.x_grainsize), -(-self.height // self.y_grainsize))

    def tile_count(self):
        c, r = self.tile_dimensions()
        return c * r

    def tile(self, idx):
        if idx >= self.tile_count():
            return None
        cols, _ = self.tile_dimensions()
        x, y = idx % cols, idx // cols
        return Tile(self.offset_x + x * self.x_grainsize,
                    self.offset_x + min(self.width, (x + 1) * self.x_grainsize),
                    self.offset_y + y * self.y_grainsize,
                    self.offset_y + min(self.height, (y + 1) * self.y_grainsize))

    def __iter__(self):
        return (self.tile(i) for i in range(self.tile_count()))


@dataclass
class TileMsg:
    tile: Tile
    data: np.ndarray  # [len(tile), 15]: Normal, Position, Albedo, Color (3 each), Variance, Z, RayDepth


class Comm:
    """One rank of a multi-GPU job (rtmi_comm): RCCL between the GPUs of a node.  Rank 0 makes the id with
    Comm.unique_id() and passes it to the other ranks; every rank then constructs its Comm at the same time."""

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        check(_lib.load().rtmi_comm_unique_id(buf))
        return buf.raw

    def __init__(self, device: Device, unique_id: bytes, rank: int, world: int):
        if len(unique_id) != _lib.COMM_ID_BYTES:
            raise ValueError("a communicator id has %d bytes" % _lib.COMM_ID_BYTES)
        self.device, self.rank, self.world = device, int(rank), int(world)
        self._l = device._l
        h = C.c_void_p()
        check(self._l.rtmi_comm_create(device._h, unique_id, self.rank, self.world, C.byref(h)))
        self._h = h

    def gather_abort(self):
        """rtmi_comm_gather_abort: this rank's turn in a frame's gather when it has nothing to bring (its render failed);
        the peers' Film.gather then raises instead of waiting."""
        check(self._l.rtmi_comm_gather_abort(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._l.rtmi_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tile_partition(n_tiles: int, world: int, rank: int) -> range:
    """rtmi_tile_partition: the tiles rank `rank` of `world` renders (interleaved)."""
    first, stride, count = C.c_uint32(), C.c_uint32(), C.c_uint32()
    check(_lib.load().rtmi_tile_partition(int(n_tiles), int(world), int(rank), C.byref(first), C.byref(stride),
                                          C.byref(count)))
    return range(first.value, first.value + stride.value * count.value, stride.value)


class Film:
    """Per-pixel accumulators on the GPU for an explicit pixel list (Dispatcher::tiles_data, executor.rs:194-201)."""

    def __init__(self, device: Device, xy, accum_device_ptr: int = 0):
        self.device = device
        self._l = device._l
        self.xy = np.ascontiguousarray(xy, dtype=np.uint32).reshape(-1, 2)
        h = C.c_void_p()
        check(self._l.rtmi_film_create(device._h, self.xy.ctypes.data, self.xy.shape[0],
                                       C.c_void_p(accum_device_ptr) if accum_device_ptr else None, C.byref(h)))
        self._h = h

    @classmethod
    def _adopt(cls, device: Device, handle):
        self = cls.__new__(cls)
        self.device, self._l, self._h = device, device._l, handle
        n = int(self._l.rtmi_film_pixel_count(handle))
        self.xy = np.zeros((n, 2), dtype=np.uint32)
        check(self._l.rtmi_film_pixels(handle, self.xy.ctypes.data))
        return self

    @classmethod
    def for_tiles(cls, device: Device, tiler: "Tiler", tiles: range, block=(8, 8)):
        """rtmi_film_create_tiles: the film of one rank's share of the tiles (`tiles` = first, first + stride, ...)."""
        if tiler.x_grainsize != tiler.y_grainsize:
            raise ValueError("square tiles only")
        tiles = tiles if isinstance(tiles, range) else range(*tiles)
        if tiles.step <= 0 or (len(tiles) and tiles.stop < tiler.tile_count()):
            raise ValueError("tiles must be a forward range running to the last tile")
        bw, bh = block if block else (0, 0)
        h = C.c_void_p()
        check(device._l.rtmi_film_create_tiles(device._h, tiler.offset_x, tiler.offset_y, tiler.width, tiler.height,
                                               tiler.x_grainsize, tiles.start, tiles.step, bw, bh, C.byref(h)))
        return cls._adopt(device, h)

    @classmethod
    def frame(cls, device: Device, width: int, height: int, offset=(0, 0)):
        """rtmi_film_create_frame: every pixel of the window, row-major -- the destination of gather()."""
        h = C.c_void_p()
        check(device._l.rtmi_film_create_frame(device._h, int(offset[0]), int(offset[1]), int(width), int(height),
                                               C.byref(h)))
        return cls._adopt(device, h)

    def gather(self, comm: Optional["Comm"], frame: Optional["Film"], root: int = 0):
        """rtmi_film_gather: this rank's accumulators into `frame` on rank `root` (collective over `comm`; comm=None is
        the single-GPU case)."""
        check(self._l.rtmi_film_gather(self._h, comm._h if comm is not None else None, int(root),
                                       frame._h if frame is not None else None))

    @property
    def n_pixels(self):
        return self.xy.shape[0]

    def close(self):
        if getattr(self, "_h", None):
            self._l.rtmi_film_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        check(self._l.rtmi_film_clear(self._h))

    def accum_device(self) -> int:
        return int(self._l.rtmi_film_accum_device(self._h) or 0)

    def download(self):
        out = np.zeros((ACCUM_PLANES, self.n_pixels), dtype=np.float32)
        check(self._l.rtmi_film_download(self._h, out.ctypes.data))
        return out

    def resolve(self, srgb=False):
        out = np.zeros((RESOLVE_PLANES, self.n_pixels), dtype=np.float32)
        check(self._l.rtmi_film_resolve(self._h, out.ctypes.data, 1 if srgb else 0))
        return out


SAMPLERS = {"stratified": 0x00, "uniform": 0x01, "pixel-centre": 0x02}
FILTERS = {"box": 0x00, "triangle": 0x10, "none": 0x20}


def render(world: CommittedHipScene, film: Film, camera: Camera, spp: int, sample_range=None, max_depth: int = 64,
           integrator: int = 0, seed: int = 0, samples_per_pass: int = 0, allowed_error: float = 0.0,
           sampler: str = "stratified", pixel_filter: str = "box"):
    """rtmi_render: adds samples `sample_range` (default 0..spp) of every film pixel.  Returns the stats dict.
    `sampler` is "stratified" (what the executor builds), "uniform" or "pixel-centre" (sampler.rs:18-112);
    `pixel_filter` is "box", "triangle" or "none" (filter.rs:11-55, radius 0.7)."""
    p = _lib.RenderParams()
    p.spp = int(spp)
    rng = sample_range if sample_range is not None else range(0, spp)
    p.sample_begin, p.sample_end = int(rng.start), int(rng.stop)
    p.max_depth = int(max_depth)
    p.integrator = int(integrator)
    p.samples_per_pass = int(samples_per_pass)
    p.seed = int(seed)
    p.allowed_error = float(allowed_error or 0.0)
    p.flags = SAMPLERS[sampler] | FILTERS[pixel_filter]
    cam = camera_pod(camera)
    st = _lib.RenderStats()
    check(world._l.rtmi_render(world._h, film._h, C.byref(cam), C.byref(p), C.byref(st)))
    return {k: getattr(st, k) for k, _ in st._fields_}


class Executor:
    """crates/rt-cli/src/executor.rs:46-58.  `integrator` is "path-tracer" or "basic"; `max_depth` rides along with it
    (Box<dyn Integrator> from_args, utils/mod.rs:147-155)."""

    def __init__(self, dimension, tile_size=32, allowed_error=None, integrator="path-tracer", camera=None, spp=32,
                 seed=0, max_depth=64):
        self.dimension = tuple(dimension)
        self.tile_size = int(tile_size)
        self.allowed_error = allowed_error
        self.integrator = integrator
        self.camera = camera if camera is not None else Camera.from_dimensions(*self.dimension)
        self.spp = int(spp)
        self.seed = int(seed)
        self.max_depth = int(max_depth)
        self.last_stats = None

    def _tiler(self, pixel_range: RenderRange):
        return Tiler(pixel_range.x.start, pixel_range.y.start, len(pixel_range.x), len(pixel_range.y), self.tile_size,
                     self.tile_size)

    def run_multithreaded(self, world, on_tile_rendered, pixel_range=None, sample_range=None, tiles=None,
                          batch_size=32, srgb=False):
        """Renders `sample_range` of every pixel in `pixel_range` on the GPU and reports each tile to
        `on_tile_rendered(TileMsg)` after every batch of `batch_size` samples (SampleCounter, executor.rs:132,333-370).
        `tiles` restricts the work to a subset of tile indices (how ranks split a frame).  `srgb` delivers the four RGB
        channels sRGB-encoded, as OutputBuffers::convert stores them (crates/rt-cli/src/output/mod.rs:46-51).
        --allowed-error is applied per batch, as the reference does (executor.rs:132,244-249)."""
        W, H = self.dimension
        pixel_range = pixel_range or RenderRange(range(0, W), range(0, H))
        sample_range = sample_range if sample_range is not None else range(0, self.spp)
        tiler = self._tiler(pixel_range)
        idxs = list(range(tiler.tile_count())) if tiles is None else list(tiles)
        tile_list = [tiler.tile(i) for i in idxs]
        xy = np.concatenate([t.pixels() for t in tile_list], 0) if tile_list else np.zeros((0, 2), np.uint32)
        offsets = np.cumsum([0] + [len(t) for t in tile_list])
        film = Film(world.scene.device, xy)
        stats = {"rays": 0, "paths": 0, "passes": 0, "gpu_ms": 0.0, "bounce_ms": 0.0, "max_bounces": 0,
                 "bounce_launches": 0, "extend_ms": 0.0, "tail_rays": 0}
        try:
            cur = sample_range.start
            while cur < sample_range.stop:
                n = min(batch_size, sample_range.stop - cur) if batch_size else sample_range.stop - cur
                st = render(world, film, self.camera, self.spp, range(cur, cur + n), self.max_depth,
                            INTEGRATORS[self.integrator], self.seed, 0, self.allowed_error or 0.0)
                for k in ("rays", "paths", "passes", "gpu_ms", "bounce_ms", "bounce_launches", "extend_ms", "tail_rays"):
                    stats[k] += st[k]
                stats["max_bounces"] = max(stats["max_bounces"], st["max_bounces"])
                cur += n
                if on_tile_rendered is not None:
                    res = film.resolve(srgb=srgb)
                    for t, a, b in zip(tile_list, offsets[:-1], offsets[1:]):
                        on_tile_rendered(TileMsg(t, res[:, a:b].T.copy()))
            self.last_stats = stats
            self.last_accum = film.download()
            self.last_xy = xy
        finally:
            film.close()
        return stats

    # one device queue does the work either way; kept for interface parity (executor.rs:148-175)
    run_monothreaded = run_multithreaded
