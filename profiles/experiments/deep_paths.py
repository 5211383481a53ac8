"""Which paths of the headline workload are the long ones, and can they be told apart when the tail kernel takes over?

The tail kernel lasts as long as its deepest path takes after the last path has been started; paths started early
finish inside the busy part of the launch.  For every sampled path alive at bounce D this script records whether the
ray entering bounce D runs inside the glass (its next hit is a back face of the closed mesh) and how many bounces the
path has in all (host build of the device code, tests/hostsim).
Usage: python profiles/experiments/deep_paths.py [levels] [pixel stride] [D]
"""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from raytracing_amd import scenes, types  # noqa: E402
from tests.hostsim import hostsim as HS  # noqa: E402

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 5
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 61
D = int(sys.argv[3]) if len(sys.argv) > 3 else 5
L = HS.lib()
L.hs_path_rays.restype = C.c_size_t
L.hs_path_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t]

t0 = time.time()
sh = HS.HostSimScene()
scenes.DragonScene.insert_into(sh, levels=levels)
sh.commit()
print("scene", sh.info["n_prims"], "primitives, %.0f s" % (time.time() - t0), flush=True)
W, H, SPP = 1920, 1080, 100
cam = types.Camera.from_dimensions(W, H)
camv = HS.camera_view(cam)
rows = []
ray = np.zeros(1, dtype=HS.RAY_DTYPE)
xy = np.zeros((1, 2), dtype=np.uint32)
for y in range(stride // 2, H, stride):
    for x in range(stride // 2, W, stride):
        xy[0] = (x, y)
        for s in range(0, SPP, 4):
            prm = HS.params(SPP, sample_begin=s, sample_end=s + 1)
            n = L.hs_path_rays(sh._h, C.byref(camv), C.byref(prm), xy.ctypes.data, 1, D, ray.ctypes.data, 1)
            if n == 0:
                continue
            _, length = sh.render_sample(cam, prm, x, y, s)
            r = ray.copy()
            r["tnear"], r["tfar"] = 1e-5, np.inf
            h = sh.intersect(r)[0]
            inside = bool(h["geom_id"] != 0xFFFFFFFF and float(np.dot(h["n"], r["dir"][0])) > 0.0)
            rows.append((length, inside))
rows = np.array(rows, dtype=np.int64)
length, inside = rows[:, 0], rows[:, 1].astype(bool)
print("%d paths alive at bounce %d, %.1f %% of them inside; %.0f s" % (len(rows), D, 100 * inside.mean(), time.time() - t0))
for cut in (8, 12, 16, 24, 32, 48, 64):
    deep = length >= cut
    if deep.sum() == 0:
        continue
    print("paths of >= %2d rays: %6d (%.3f %% of those alive), %.1f %% of them inside at bounce %d" % (
        cut, deep.sum(), 100 * deep.mean(), 100 * inside[deep].mean(), D))
print("mean rays still to trace: inside %.2f, outside %.2f" % ((length[inside] - D).mean(), (length[~inside] - D).mean()))
