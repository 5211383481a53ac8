"""Host model of a wave-shared queue of deferred record tests for the traversal kernel, against today's scheduling
(majority phase voting), on camera and bounce rays of the dragon stand-in.  Costs in ns of SIMD time per wave step."""
import sys, ctypes as C; sys.path.insert(0, '.')
import numpy as np
from oracle import oracle as O
from raytracing_amd import scenes, types
from tests.hostsim import hostsim as HS
from tests.helpers import full_frame_xy
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 3
L = HS.lib()
L.hs_wave_sim.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
L.hs_queue_sim.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
sh = HS.HostSimScene(); scenes.DragonScene.insert_into(sh, levels=levels); sh.commit(); print(sh.info)
W, H = 480, 270
cam = types.Camera.from_dimensions(W, H); xy = full_frame_xy(W, H)
so = O.OracleScene()  # only for camera rays
rays = so.camera_rays(cam, O.make_params(spp=4), xy[::2], 1); rays["tnear"] = 1e-5; rays["tfar"] = np.inf
rng = np.random.default_rng(3)
def bounce(r):
    h = sh.intersect(r); hit = h["geom_id"] != 0xFFFFFFFF
    b = np.zeros(int(hit.sum()), dtype=HS.RAY_DTYPE); b["org"] = h["pos"][hit]
    d = rng.normal(size=(b.shape[0], 3)).astype(np.float32); b["dir"] = d / np.sqrt((d * d).sum(1))[:, None]; b["tnear"] = 1e-5; b["tfar"] = np.inf
    return b
sets = {"primary": rays}
b1 = bounce(rays); sets["bounce1"] = b1; b2 = bounce(b1); sets["bounce2"] = b2
def run(fn, *a):
    out = (C.c_double * 8)(); fn(*a, out); return list(out)
for name, r in sets.items():
    r = np.ascontiguousarray(r); n = r.shape[0]
    cost = (C.c_double * 4)(240, 350, 250, 20)
    base = run(L.hs_wave_sim, sh._h, r.ctypes.data, n, 0, 48, 0, 0, cost)
    print(name, n, "rays; today: ns/ray %.0f util %.3f node lanes/iter %.1f record lanes/iter %.1f nodes/ray %.2f records/ray %.2f" % (
        base[0] / n / 64, base[1] / base[0], base[2] / max(base[4], 1), base[3] / max(base[5], 1), base[2] / n, base[3] / n))
    for rb, fire, low in ((48, 64, 16), (56, 64, 16), (48, 64, 32), (56, 64, 8), (60, 64, 24)):
            for post, leaf in ((25, 390), (40, 420)):
                c5 = (C.c_double * 5)(245, leaf, 250, 20, post)
                o = run(L.hs_queue_sim, sh._h, r.ctypes.data, n, rb, fire, 128, low, c5)
                print("   queue refill<%d fire %d low %d post %d record %d: ns/ray %.0f (x%.2f) util %.3f node lanes/iter %.1f record lanes/iter %.1f nodes/ray %.2f records/ray %.2f partial record phases %.2f" % (
                    rb, fire, low, post, leaf, o[0] / n / 64, base[0] / o[0], o[1] / o[0], o[2] / max(o[4], 1), o[3] / max(o[5], 1), o[2] / n, o[3] / n, o[7] / max(o[5], 1)))
