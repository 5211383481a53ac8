"""Spatial splits in the BVH builder: tree size and visits per ray with and without them (host build of the device
code, tests/hostsim), on the Cornell box, the bunny and the dragon stand-in.  Rays: camera rays of a coarse grid of
pixels, then two generations of bounce rays (uniform random directions from the hit points); for the two synthetic
meshes of tests/helpers.py (small triangles with a few long slanted ones among them; a soup of needles) random rays
through their cube instead of camera rays, and the same for a soup of 200 000 triangles of mixed sizes.  RTMI_SPATIAL_GAIN, RTMI_SPATIAL_ALPHA, RTMI_SPATIAL_LARGE in the
environment move the builder's thresholds (csrc/bvh.cpp).
Usage: python profiles/experiments/spatial_splits.py [dragon levels=5]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from oracle import oracle as O  # noqa: E402  (camera rays only)
from raytracing_amd import scenes, types  # noqa: E402
from tests.hostsim import hostsim as HS  # noqa: E402

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 5
LEAF = 0x80000000
L = HS.lib()
L.hs_trace.restype = C.c_size_t
L.hs_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
L.hs_items.restype = C.c_size_t
L.hs_items.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]

W, H = 480, 270
cam = types.Camera.from_dimensions(W, H)
xy = np.array([(x, y) for y in range(0, H, 2) for x in range(0, W, 2)], dtype=np.uint32)
so = O.OracleScene()
cam_rays = so.camera_rays(cam, O.make_params(spp=4), xy, 1)
cam_rays["tnear"] = 1e-5
cam_rays["tfar"] = np.inf


def visits(sh, rays):
    rays = np.ascontiguousarray(rays, dtype=HS.RAY_DTYPE)
    n = rays.shape[0]
    offs = np.zeros(n + 1, dtype=np.uint64)
    total = L.hs_trace(sh._h, rays.ctypes.data, n, None, 0, offs.ctypes.data)
    buf = np.zeros(total, dtype=np.uint32)
    L.hs_trace(sh._h, rays.ctypes.data, n, buf.ctypes.data, total, offs.ctypes.data)
    rec = int(((buf & LEAF) != 0).sum())
    return (total - rec) / n, rec / n


def bounce(sh, rays, rng):
    h = sh.intersect(rays)
    hit = h["geom_id"] != 0xFFFFFFFF
    b = np.zeros(int(hit.sum()), dtype=HS.RAY_DTYPE)
    b["org"] = h["pos"][hit]
    d = rng.normal(size=(b.shape[0], 3)).astype(np.float32)
    b["dir"] = d / np.sqrt((d * d).sum(1))[:, None]
    b["tnear"] = 1e-5
    b["tfar"] = np.inf
    return b, h


def insert_mixed(s):
    from tests.helpers import mixed_scale_mesh
    s.insert_mesh(0, *mixed_scale_mesh())


def insert_needles(s):
    from tests.helpers import needle_mesh
    s.insert_mesh(0, *needle_mesh(20000))


def insert_soup(s, n=200000):
    """Triangles of sizes between 0.003 and 2 thrown into a cube of 20: large and small ones overlap everywhere."""
    rng = np.random.default_rng(5)
    c = rng.uniform(-10, 10, (n, 3))
    size = 10.0 ** rng.uniform(-2.5, 0.3, (n, 1, 1))
    v = (c[:, None, :] + rng.normal(size=(n, 3, 3)) * size).reshape(-1, 3).astype(np.float32)
    s.insert_mesh(0, v, np.arange(3 * n, dtype=np.uint32).reshape(-1, 3))


def cube(sh, rng, n=20000, half=8.0):
    r = np.zeros(n, dtype=HS.RAY_DTYPE)
    r["org"] = rng.uniform(-half, half, (n, 3))
    d = rng.normal(size=(n, 3)).astype(np.float32)
    r["dir"] = d / np.sqrt((d * d).sum(1))[:, None]
    r["tnear"] = 1e-5
    r["tfar"] = np.inf
    return r


for name, insert in (("mixed-scale mesh", insert_mixed), ("soup of 200000", insert_soup), ("20000 needles", insert_needles),
                     ("cornell", lambda s: scenes.CornellBoxScene.insert_into(s)),
                     ("bunny", lambda s: scenes.StandfordBunnyScene.insert_into(s)),
                     ("dragon stand-in", lambda s: scenes.DragonScene.insert_into(s, levels=levels))):
    ref_hits = None
    for spatial in ("0", "1"):
        os.environ["RTMI_SPATIAL"] = spatial
        sh = HS.HostSimScene()
        insert(sh)
        t0 = time.time()
        sh.commit()
        dt = time.time() - t0
        n_items = L.hs_items(sh._h, None, 0)
        rng = np.random.default_rng(7)
        first = cam_rays if name in ("cornell", "bunny", "dragon stand-in") else cube(sh, np.random.default_rng(1))
        sets = [("camera" if first is cam_rays else "random", first)]
        b1, h0 = bounce(sh, first, rng)
        b2, h1 = bounce(sh, b1, rng)
        sets += [("bounce 1", b1), ("bounce 2", b2)]
        if ref_hits is None:
            ref_hits = (h0, h1)
        else:
            for a, b in zip(ref_hits, (h0, h1)):
                assert a.tobytes() == b.tobytes(), "hits differ between the two trees"
        line = "%-16s spatial=%s build %.2f s, %d nodes, %d records (%d cuts), %d items, stack %d:" % (
            name, spatial, dt, sh.info["n_nodes"], sh.info["n_records"], sh.info["spatial_splits"], n_items, sh.info["stack_need"])
        for label, r in sets:
            nv, rv = visits(sh, r)
            line += "  %s %.2f nodes %.2f records" % (label, nv, rv)
        print(line, flush=True)
        sh.close() if hasattr(sh, "close") else None
os.environ.pop("RTMI_SPATIAL", None)
