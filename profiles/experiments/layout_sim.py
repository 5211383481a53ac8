"""How many 128-byte lines does a ray fetch from beyond L2 under different memory layouts of the same tree?

Traces the items (BVH4 nodes, primitive records; 64 bytes each) that real bounce rays of the headline workload visit
(host build of the device code, tests/hostsim), then prices each layout: an item costs a fabric line read unless its
line is hot (visited by more than 1 ray in HOT_ONE_IN: such lines stay in a 4 MB L2 between two visits) or the ray has
already fetched that line.  Layouts:
  current   nodes breadth-first in one array, records in leaf order in another (siblings share lines by accident of parity)
  pairs     one array of 128-byte pairs: a node with its largest child (node or record); leftover records two by two
Usage: python profiles/experiments/layout_sim.py [levels] [pixel stride]

The script reads the tree as it was laid out when the study was made (nodes breadth-first, then the records, four
child references per node); the builder has since moved to neighbouring children named by one index, so it needs the
sources of that time (the commit that added it) to run.
"""
import ctypes as C
import os
import sys
import time

os.environ.pop("RTMI_LAYOUT", None)  # the tree in its breadth-first placement: item index = node index

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from raytracing_amd import scenes, types  # noqa: E402
from tests.hostsim import hostsim as HS  # noqa: E402

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 5
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 31
HOT_ONE_IN = 1600
EMPTY, LEAF = 0xFFFFFFFF, 0x80000000

L = HS.lib()
L.hs_trace.restype = C.c_size_t
L.hs_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
L.hs_items.restype = C.c_size_t
L.hs_items.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
L.hs_path_rays.restype = C.c_size_t
L.hs_path_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t]

t0 = time.time()
sh = HS.HostSimScene()
scenes.DragonScene.insert_into(sh, levels=levels)
sh.commit()
print("scene", sh.info, "%.0f s" % (time.time() - t0), flush=True)
n_nodes = sh.info["n_nodes"]
n_items = L.hs_items(sh._h, None, 0)
nodes = np.zeros((n_items, 16), dtype=np.uint32)
L.hs_items(sh._h, nodes.ctypes.data, n_items)
nodes = nodes[:n_nodes]
rec_base = (n_nodes + 1) // 2 * 2
refs = nodes[:, 12:16].copy()
is_leaf = (refs & LEAF) != 0
valid = refs != EMPTY
is_leaf &= valid
refs[is_leaf] -= rec_base  # leaf references as record slots
slots = (refs & 0x0FFFFFFF)
n_rec = int(slots[is_leaf].max()) + 1
print("nodes %d records %d" % (n_nodes, n_rec))

# child areas from the quantised boxes
step = nodes[:, 3:6].view(np.float32)
q = nodes[:, 6:12]
ext = np.zeros((n_nodes, 4, 3), dtype=np.float64)
for k in range(4):
    for a in range(3):
        lo = (q[:, a] >> (8 * k)) & 255
        hi = (q[:, 3 + a] >> (8 * k)) & 255
        ext[:, k, a] = (hi.astype(np.float64) - lo) * step[:, a]
area = ext[:, :, 0] * ext[:, :, 1] + ext[:, :, 1] * ext[:, :, 2] + ext[:, :, 2] * ext[:, :, 0]
area[~valid] = -1.0


def layout_current():
    """item -> 64-byte slot address, in two separate arrays (records offset so that lines never alias nodes)."""
    node_addr = np.arange(n_nodes, dtype=np.int64)
    rec_addr = np.arange(n_rec, dtype=np.int64) + ((n_nodes + 1) // 2 * 2)
    return node_addr, rec_addr


def layout_pairs(fav):
    """Unified array of pairs.  fav[n] = child slot (0..3) of node n placed in the same line as n."""
    node_addr = np.full(n_nodes, -1, dtype=np.int64)
    rec_addr = np.full(n_rec, -1, dtype=np.int64)
    refs_l = refs.tolist()
    fav_l = fav.tolist()
    nxt = 0
    from collections import deque
    heads = deque([("n", 0)])
    while heads:
        kind, idx = heads.popleft()
        if kind == "r":  # a pair (or a single) of leftover records
            for j, s in enumerate(idx):
                rec_addr[s] = nxt + j
            nxt += 2
            continue
        node_addr[idx] = nxt
        kids = refs_l[idx]
        f = fav_l[idx]
        tail = kids[f]
        pending_nodes, pending_recs = [], []
        if tail & LEAF:
            rec_addr[tail & 0x0FFFFFFF] = nxt + 1
        else:
            node_addr[tail] = nxt + 1
            for c in refs_l[tail]:
                if c == EMPTY:
                    continue
                (pending_recs if c & LEAF else pending_nodes).append(c)
        nxt += 2
        for k, c in enumerate(kids):
            if c == EMPTY or k == f:
                continue
            (pending_recs if c & LEAF else pending_nodes).append(c)
        for c in pending_nodes:
            heads.append(("n", c))
        # leftover records: the tail's own children first (they were appended first), two by two
        recs = [c & 0x0FFFFFFF for c in pending_recs]
        for i in range(0, len(recs), 2):
            heads.append(("r", recs[i:i + 2]))
    assert (node_addr >= 0).all() and (rec_addr >= 0).all()
    return node_addr, rec_addr, nxt


cam = types.Camera.from_dimensions(1920, 1080)
ys, xs = np.mgrid[stride // 2:1080:stride, stride // 2:1920:stride]
xy = np.ascontiguousarray(np.stack([xs.ravel(), ys.ravel()], 1).astype(np.uint32))
camv = HS.CameraView()
camv.width, camv.height = 1920, 1080
camv.half_width, camv.half_height = cam.viewport_half_width, cam.viewport_half_height
camv.focal_length, camv.aperture = cam.focal_length, cam.aperture
camv.lens = (C.c_float * 3)(*cam.center_of_lens)
camv.rot = (C.c_float * 4)(*cam.rotation)
prm = HS.Params()
prm.spp, prm.sample_begin, prm.sample_end, prm.max_depth, prm.integrator, prm.seed, prm.flags = 100, 0, 24, 64, 0, 0, 0

traces = {}
origins = {}
for depth in (1, 2, 3, 5, 8):
    t0 = time.time()
    cap = xy.shape[0] * 24
    rays = np.zeros(cap, dtype=HS.RAY_DTYPE)
    n = L.hs_path_rays(sh._h, C.byref(camv), C.byref(prm), xy.ctypes.data, xy.shape[0], depth, rays.ctypes.data, cap)
    rays = np.ascontiguousarray(rays[:n])
    offsets = np.zeros(n + 1, dtype=np.uint64)
    total = L.hs_trace(sh._h, rays.ctypes.data, n, None, 0, offsets.ctypes.data)
    items = np.zeros(total, dtype=np.uint32)
    L.hs_trace(sh._h, rays.ctypes.data, n, items.ctypes.data, total, offsets.ctypes.data)
    leaf = (items & LEAF) != 0
    items[leaf] -= rec_base
    traces[depth] = (items, offsets.astype(np.int64))
    origins[depth] = rays["org"].copy()
    print("depth %d: %d rays, %.1f items per ray (%.0f s)" % (depth, n, total / max(n, 1), time.time() - t0), flush=True)


def octant_groups(org):
    """Eight equally filled groups of rays by origin: median splits along x, then y, then z."""
    g = np.zeros(org.shape[0], dtype=np.int64)
    for a in range(3):
        for cell in np.unique(g):
            m = g == cell
            med = np.median(org[m, a])
            g[m] = g[m] * 2 + (org[m, a] > med)
    return g


def evaluate(name, node_addr, rec_addr, by_origin=False):
    for depth, (items, offsets) in traces.items():
        n = offsets.shape[0] - 1
        if by_origin:
            # every L2 (one per XCD) sees the rays of one origin cell only
            grp = octant_groups(origins[depth])
            leaf = (items & LEAF) != 0
            addr = np.where(leaf, rec_addr[np.where(leaf, items & 0x0FFFFFFF, 0)], node_addr[np.where(leaf, 0, items)])
            line = addr >> 1
            ray = np.repeat(np.arange(n, dtype=np.int64), np.diff(offsets))
            uniq = np.unique(ray * (1 << 34) + line)
            uray, uline = uniq >> 34, uniq & ((1 << 34) - 1)
            cold_total = 0
            for c in range(8):
                m = grp[uray] == c
                lines, counts = np.unique(uline[m], return_counts=True)
                hot = lines[counts > (grp == c).sum() / HOT_ONE_IN]
                cold_total += (~np.isin(uline[m], hot)).sum()
            print("  %-8s depth %d, rays dealt to the 8 L2s by origin cell: cold lines/ray %.2f" % (name, depth, cold_total / n))
            continue
        leaf = (items & LEAF) != 0
        addr = np.where(leaf, rec_addr[np.where(leaf, items & 0x0FFFFFFF, 0)], node_addr[np.where(leaf, 0, items)])
        line = addr >> 1
        ray = np.repeat(np.arange(n, dtype=np.int64), np.diff(offsets))
        # distinct lines per ray
        key = ray * (1 << 34) + line
        uniq = np.unique(key)
        uline = uniq & ((1 << 34) - 1)
        # hot lines: touched by more than one ray in HOT_ONE_IN
        lines, counts = np.unique(uline, return_counts=True)
        hot = lines[counts > n / HOT_ONE_IN]
        cold = ~np.isin(uline, hot)
        print("  %-8s depth %d: items/ray %.2f  distinct lines/ray %.2f  cold lines/ray %.2f  (hot lines %d = %.1f MB)" % (
            name, depth, items.shape[0] / n, uniq.shape[0] / n, cold.sum() / n, hot.shape[0], hot.shape[0] * 128 / 1e6))


na, ra = layout_current()
evaluate("current", na, ra)
evaluate("current", na, ra, by_origin=True)
fav_area = np.argmax(area, 1)
na, ra, used = layout_pairs(fav_area)
print("pairs (largest child): %d slots for %d items (%.1f %% padding)" % (used, n_nodes + n_rec, 100.0 * (used - n_nodes - n_rec) / used))
evaluate("pairs", na, ra)
evaluate("pairs", na, ra, by_origin=True)
# upper bound: the child the traced rays visit most often
visits_n = np.zeros(n_nodes, dtype=np.int64)
visits_r = np.zeros(n_rec, dtype=np.int64)
for items, _ in traces.values():
    leaf = (items & LEAF) != 0
    np.add.at(visits_n, items[~leaf], 1)
    np.add.at(visits_r, items[leaf] & 0x0FFFFFFF, 1)
cv = np.where(is_leaf, visits_r[np.where(is_leaf, slots, 0)], visits_n[np.where(valid & ~is_leaf, refs, 0)]).astype(np.float64)
cv[~valid] = -1.0
na, ra, used = layout_pairs(np.argmax(cv + 1e-9 * np.maximum(area, 0), 1))
evaluate("pairs*", na, ra)
print("(pairs* = the child these very rays visit most: an upper bound no builder has)")
