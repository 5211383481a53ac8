"""How many 128-byte lines would a ray fetch less if the four children of a node were paired into its two lines
differently?  The children of a node are neighbours in the item array (child k = item first + k, two items per line), so
a ray that visits children 0 and 1 fetches one line and a ray that visits 0 and 2 fetches two.  For real bounce rays of
the dragon stand-in (host build of the device code) this counts, per node, the lines its children cost under the three
possible pairings ({01|23} as built, {02|13}, {03|12}) and reports the total for the builder's order, for the best
pairing per node chosen WITH HINDSIGHT on the same rays (an upper bound no build-time rule can beat), and for pairing
by box overlap.  Every item a ray visits is charged (no cache model): the question is only how much the order can move.
Usage: python profiles/experiments/sibling_lines.py [levels=5] [pixel stride=29]"""
import ctypes as C
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from raytracing_amd import scenes, types  # noqa: E402
from tests.hostsim import hostsim as HS  # noqa: E402

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 5
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 29
LEAF = 0x80000000
L = HS.lib()
L.hs_path_rays.restype = C.c_size_t
L.hs_path_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t]

sh = HS.HostSimScene()
scenes.DragonScene.insert_into(sh, levels=levels)
sh.commit()
print("scene", sh.info, flush=True)
items = sh.items()
W, H = 1920, 1080
cam = HS.camera_view(types.Camera.from_dimensions(W, H))
xy = np.array([(x, y) for y in range(0, H, stride) for x in range(0, W, stride)], dtype=np.uint32)
prm = HS.params(100, 0, 4)

# parent group of every item: group id = first item of the group; slot = item - first
n_items = items.shape[0]
group = np.full(n_items, -1, dtype=np.int64)
slot = np.zeros(n_items, dtype=np.int64)
is_node = np.zeros(n_items, dtype=bool)
is_node[0] = True
frontier = np.array([0], dtype=np.int64)
while frontier.size:
    meta, first = items[frontier, 10], items[frontier, 11].astype(np.int64)
    count = ((meta >> 24) & 3).astype(np.int64) + 1
    leaf_bits = (meta >> 16) & 15
    nxt = []
    for k in range(4):
        sel = count > k
        group[first[sel] + k] = first[sel]
        slot[first[sel] + k] = k
        inner = sel & (((leaf_bits >> k) & 1) == 0)
        nxt.append(first[inner] + k)
    frontier = np.concatenate(nxt)
    is_node[frontier] = True

for depth in (2, 5):
    cap = xy.shape[0] * 4
    rays = np.zeros(cap, dtype=HS.RAY_DTYPE)
    n = L.hs_path_rays(sh._h, C.byref(cam), C.byref(prm), xy.ctypes.data, xy.shape[0], depth, rays.ctypes.data, cap)
    rays = rays[:n]
    offs = np.zeros(n + 1, dtype=np.uint64)
    total = L.hs_trace(sh._h, rays.ctypes.data, n, None, 0, offs.ctypes.data)
    buf = np.zeros(total, dtype=np.uint32)
    L.hs_trace(sh._h, rays.ctypes.data, n, buf.ctypes.data, total, offs.ctypes.data)
    it = (buf & 0x0FFFFFFF).astype(np.int64)
    ray = np.repeat(np.arange(n), np.diff(offs).astype(np.int64))
    keep = it != 0  # the root has no group
    it, ray = it[keep], ray[keep]
    g, s = group[it], slot[it]
    # per (ray, group): bit mask of visited slots
    key = ray * n_items + g
    order = np.argsort(key, kind="stable")
    key, bits = key[order], (1 << s[order])
    uniq, start = np.unique(key, return_index=True)
    mask = np.bitwise_or.reduceat(bits, start)
    grp = uniq % n_items
    # lines a visit mask costs under each pairing
    def lines(mask, pairs):
        a = ((mask >> pairs[0][0]) | (mask >> pairs[0][1])) & 1
        b = ((mask >> pairs[1][0]) | (mask >> pairs[1][1])) & 1
        return a + b
    pairings = [((0, 1), (2, 3)), ((0, 2), (1, 3)), ((0, 3), (1, 2))]
    cost = np.stack([lines(mask, p) for p in pairings], 1)  # [visits, 3]
    per_group = np.zeros((n_items, 3), dtype=np.int64)
    np.add.at(per_group, grp, cost)
    built = per_group[:, 0].sum()
    best = per_group.min(1).sum()
    root_line = n  # every ray fetches the root
    print("bounce %d: %d rays, %.2f items per ray; sibling-group lines per ray: as built %.2f, best pairing per node with hindsight %.2f (%.1f %% fewer)"
          % (depth, n, (total) / n, built / n, best / n, 100.0 * (built - best) / built), flush=True)
