"""Host model of wave scheduling policies for the traversal kernel, on bounce rays of the dragon stand-in."""
import sys, time, ctypes as C; sys.path.insert(0,'.')
import numpy as np
from oracle import oracle as O
from raytracing_amd import scenes, types
from tests.hostsim import hostsim as HS
from tests.helpers import full_frame_xy
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 3
L = HS.lib()
L.hs_wave_sim.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
L.hs_block_sim.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
sh = HS.HostSimScene(); scenes.DragonScene.insert_into(sh, levels=levels); sh.commit(); print(sh.info)
W,H=480,270
cam=types.Camera.from_dimensions(W,H); xy=full_frame_xy(W,H)
so=O.OracleScene()  # only for camera rays
rays=so.camera_rays(cam,O.make_params(spp=4),xy[::2],1); rays["tnear"]=1e-5; rays["tfar"]=np.inf
rng=np.random.default_rng(3)
def bounce(r):
    h=sh.intersect(r); hit=h["geom_id"]!=0xFFFFFFFF
    b=np.zeros(int(hit.sum()),dtype=HS.RAY_DTYPE); b["org"]=h["pos"][hit]
    d=rng.normal(size=(b.shape[0],3)).astype(np.float32); b["dir"]=d/np.sqrt((d*d).sum(1))[:,None]; b["tnear"]=1e-5; b["tfar"]=np.inf
    return b
sets={"primary":rays}
b1=bounce(rays); sets["bounce1"]=b1; b2=bounce(b1); sets["bounce2"]=b2
def run(fn,*a):
    out=(C.c_double*8)(); fn(*a,out); return list(out)
for name,r in sets.items():
    r=np.ascontiguousarray(r); n=r.shape[0]
    cost=(C.c_double*4)(135,200,150,12)
    base=run(L.hs_wave_sim, sh._h, r.ctypes.data, n, 0, 48, 0, 0, cost)
    print(name, n, "rays; base: time/ray %.0f util %.3f node lanes/iter %.1f leaf lanes/iter %.1f nodes/ray %.1f leaves/ray %.1f"%(base[0]/n, base[1]/base[0], base[2]/max(base[4],1), base[3]/max(base[5],1), base[2]/n, base[3]/n))
    for nw in (4,):
        for swap in (20,40):
          for rb in (48,-16,-24,-32,-40):
                c5=(C.c_double*5)(135,200,150,16,swap)
                o=run(L.hs_block_sim, sh._h, r.ctypes.data, n, rb, nw, c5)
                print("rb",rb,end=" ")
                print("   block nw=%d swap=%d: time/ray %.0f (x%.2f) util %.3f node lanes/iter %.1f leaf %.1f swaps/iter %.2f refills %d"%(nw,swap,o[0]/n, base[0]/o[0], o[1]/o[0], o[2]/max(o[4],1), o[3]/max(o[5],1), o[7]/(o[4]+o[5]), o[6]))
