"""Throughput of the query seam (rtmi_intersect_device, the batch form of Shape::intersection_full) on the 5 M-triangle
stand-in: camera rays of one sample (coherent) and rays between random surface points (incoherent), device resident.
Usage (GPU box, repo root): python profiles/experiments/intersect_rate.py"""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from oracle import oracle as O  # camera rays only  # noqa: E402
from raytracing_amd import _lib, api, scenes, types  # noqa: E402
from tests import hip_rt  # noqa: E402
from tests.helpers import full_frame_xy  # noqa: E402

dev = api.Device(0)
sg = api.HipScene(dev)
scenes.DragonScene.insert_into(sg)
world = sg.commit().into_world()
print(world.info())
W, H = 1920, 1080
cam = types.Camera.from_dimensions(W, H)
so = O.OracleScene()
cam_rays = so.camera_rays(cam, O.make_params(spp=100), full_frame_xy(W, H), 3)
cam_rays["tnear"], cam_rays["tfar"] = 1e-5, np.inf
cam8 = np.concatenate([cam_rays] + [so.camera_rays(cam, O.make_params(spp=100), full_frame_xy(W, H), k) for k in range(4, 11)])
cam8["tnear"], cam8["tfar"] = 1e-5, np.inf
first = world.intersect(cam_rays)
hit = first["geom_id"] == 0  # the mesh
pts = first["pos"][hit]
rng = np.random.default_rng(5)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
a, b = pts[rng.integers(0, pts.shape[0], n)], pts[rng.integers(0, pts.shape[0], n)]
d = b - a
ln = np.sqrt((d * d).sum(1, dtype=np.float32))
ok = ln > 1e-4
inc = np.zeros(int(ok.sum()), dtype=_lib.RAY_DTYPE)
inc["org"], inc["dir"] = a[ok], d[ok] / ln[ok][:, None]
inc["tnear"], inc["tfar"] = 1e-5, np.inf
stream = hip_rt.Stream()
sp = C.c_void_p(stream.handle)
for name, rays, entry in (("camera rays", cam_rays, world.intersect_device), ("camera rays, 8 samples", cam8, world.intersect_device), ("surface-to-surface rays", inc, world.intersect_device),
                          ("surface-to-surface rays, static entry point", inc, world.intersect_device_static)):
    d_rays = hip_rt.DeviceBuffer.from_array(rays)
    d_hits = hip_rt.DeviceBuffer(rays.shape[0] * _lib.HIT_DTYPE.itemsize)
    entry(d_rays.ptr, d_hits.ptr, rays.shape[0], sp)
    stream.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        entry(d_rays.ptr, d_hits.ptr, rays.shape[0], sp)
    stream.synchronize()
    dt = (time.perf_counter() - t0) / reps
    hits = d_hits.download(_lib.HIT_DTYPE)
    print("%-46s %9d rays  %7.2f ms  %7.0f Mrays/s  (%.0f %% hit)" % (name, rays.shape[0], dt * 1e3, rays.shape[0] / dt / 1e6,
                                                                    100.0 * (hits["geom_id"] != 0xFFFFFFFF).mean()))
    d_rays.free()
    d_hits.free()
