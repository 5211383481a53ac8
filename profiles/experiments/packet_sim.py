"""Would a wave-packet traversal pay for the camera rays of the headline workload?

A packet variant gives the 64 camera rays of an 8 x 8 pixel block (one sample index) ONE traversal: a wave-shared
stack, every node fetched once for the wave, all lanes slab-testing the same node, a child visited when any lane hits
it.  On a SIMT machine the instruction stream of a node step costs the same whether one lane or 64 execute it, so what
decides is the number of wave steps:
  per-lane traversal (the shipped kernels): the wave steps until its slowest lane is done; with lane refill the cost is
      sum over lanes of (node visits, record tests) / lanes busy per step (measured on the GPU: 0.72 at bounce 0);
  packet traversal: one node step per node ANY lane visits, one record step per record ANY lane tests (every lane
      active, most of them on a box or a triangle their ray misses).
This script traces the items real camera rays visit (host build of the device code, tests/hostsim) for a sample of
8 x 8 blocks of the 1920 x 1080 frame and prints both counts, over all blocks and over the blocks that see the mesh,
also for a hybrid that shares only the top `k` levels of the tree and leaves the rest to the per-lane code.

Usage: python profiles/experiments/packet_sim.py [levels=5] [block stride=37]
"""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from oracle import oracle as O  # noqa: E402  (camera rays only)
from raytracing_amd import scenes, types  # noqa: E402
from tests.hostsim import hostsim as HS  # noqa: E402

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 5
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 37
LEAF = 0x80000000
UTIL = 0.72  # lanes busy per step of extend_queue_kernel at bounce 0 (profiles/r03_pmc/per_launch.txt)

L = HS.lib()
L.hs_trace.restype = C.c_size_t
L.hs_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
L.hs_items.restype = C.c_size_t
L.hs_items.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]

t0 = time.time()
sh = HS.HostSimScene()
scenes.DragonScene.insert_into(sh, levels=levels)
sh.commit()
print("scene", sh.info, "%.0f s" % (time.time() - t0), flush=True)

# depth of every inner node: children of item i are items first .. first + count - 1 (word 11; child count - 1 in bits 24-25 and leaf flags in bits 16-19 of word 10)
n_items = L.hs_items(sh._h, None, 0)
items = np.zeros((n_items, 16), dtype=np.uint32)
L.hs_items(sh._h, items.ctypes.data, n_items)
n_nodes = sh.info["n_nodes"]
depth = np.full(n_items, -1, dtype=np.int32)
depth[0] = 0
frontier = np.array([0], dtype=np.int64)
d = 0
while frontier.size:
    meta, first = items[frontier, 10], items[frontier, 11].astype(np.int64)
    count = ((meta >> 24) & 3).astype(np.int64) + 1
    leaf_bits = (meta >> 16) & 15
    nxt = []
    for k in range(4):
        sel = (count > k) & (((leaf_bits >> k) & 1) == 0)
        nxt.append(first[sel] + k)
    frontier = np.concatenate(nxt)
    d += 1
    depth[frontier] = d
print("inner-node levels: %d" % d)

W, H = 1920, 1080
cam = types.Camera.from_dimensions(W, H)
so = O.OracleScene()
prm = O.make_params(spp=100)
blocks = [(bx, by) for by in range(0, H // 8) for bx in range(0, W // 8)][::stride]
print("%d blocks of 8 x 8 pixels" % len(blocks), flush=True)

tot = {"blocks": 0, "mesh_blocks": 0}
acc = {}


def add(key, mesh, **kw):
    for scope in (("all",) + (("mesh",) if mesh else ())):
        a = acc.setdefault((key, scope), {})
        for k, v in kw.items():
            a[k] = a.get(k, 0.0) + v


cap = 1 << 22
buf = np.zeros(cap, dtype=np.uint32)
offs = np.zeros(65, dtype=np.uint64)
for bi, (bx, by) in enumerate(blocks):
    xy = np.array([(bx * 8 + i % 8, by * 8 + i // 8) for i in range(64)], dtype=np.uint32)
    rays = so.camera_rays(cam, prm, xy, bi % 100)
    rays["tnear"] = 1e-5
    rays["tfar"] = np.inf
    rays = np.ascontiguousarray(rays, dtype=HS.RAY_DTYPE)
    n = L.hs_trace(sh._h, rays.ctypes.data, 64, buf.ctypes.data, cap, offs.ctypes.data)
    assert n <= cap
    it = buf[:n]
    is_rec = (it & LEAF) != 0
    per_lane_nodes = int((~is_rec).sum())
    per_lane_recs = int(is_rec.sum())
    uniq = np.unique(it)
    u_rec = int(((uniq & LEAF) != 0).sum())
    u_nodes = uniq[(uniq & LEAF) == 0]
    mesh = per_lane_recs > 0
    tot["blocks"] += 1
    tot["mesh_blocks"] += int(mesh)
    add("full", mesh, lane_nodes=per_lane_nodes, lane_recs=per_lane_recs, pk_nodes=len(u_nodes), pk_recs=u_rec)
    node_depth = depth[it[~is_rec]]
    for k in (2, 4, 6, 8):
        # hybrid: levels < k once per wave (every node any lane visits there), deeper levels and records per lane
        shared = int((depth[u_nodes] < k).sum())
        own_top = int((node_depth < k).sum())
        add("top%d" % k, mesh, shared=shared, own_top=own_top)

print("blocks %d, of which %d see the mesh" % (tot["blocks"], tot["mesh_blocks"]))
for scope in ("all", "mesh"):
    a = acc[("full", scope)]
    nb = tot["blocks"] if scope == "all" else tot["mesh_blocks"]
    lane_steps = (a["lane_nodes"] + a["lane_recs"]) / (64 * UTIL)
    print("[%s blocks] per ray: %.1f node visits, %.2f record tests" % (scope, a["lane_nodes"] / nb / 64, a["lane_recs"] / nb / 64))
    print("   per-lane traversal: %.0f node + %.0f record wave steps per block (at %.2f lanes busy)"
          % (a["lane_nodes"] / (64 * UTIL) / nb, a["lane_recs"] / (64 * UTIL) / nb, UTIL))
    print("   packet traversal:   %.0f node + %.0f record wave steps per block  -> x%.1f the wave steps"
          % (a["pk_nodes"] / nb, a["pk_recs"] / nb, (a["pk_nodes"] + a["pk_recs"]) / lane_steps))
    for k in (2, 4, 6, 8):
        h = acc[("top%d" % k, scope)]
        saved = h["own_top"] / (64 * UTIL) - h["shared"]
        print("   hybrid, top %d levels shared: %.1f shared steps replace %.1f per-lane ones: %+.1f %% wave steps"
              % (k, h["shared"] / nb, h["own_top"] / (64 * UTIL) / nb, -100.0 * saved / lane_steps))
