"""Deterministic stand-in for the dragon mesh the reference does not distribute (`obj/dragon.obj`).

The Stanford bunny (assets/obj/standford_bunny.obj) is centred, scaled to a height of 140 units and
midpoint-subdivided `levels` times.  The height is calibrated against the reference's own render of the scene
(output/dragon100/Albedo.jpeg, 1920x1080): under the scene's transform (scale 0.01, translation (0, 0, -1), rotation
1.1 pi about Y, crates/rt/src/scene/examples/dragon.rs:27-31) and the fixed camera, the reference's dragon covers
33.8 % of the frame and leaves 57.5 % sky; at 140 units the stand-in covers 33.8 % and leaves 58.2 %, with first-hit
distances (0.71 / 0.78 / 0.88 at the 5th / 25th / 50th percentile) like the reference's Z channel shows (0.74 / 0.91 /
>= 1).  So the share of camera paths that enter the glass, which is what makes this scene expensive, is the
reference's.  Every new vertex of the subdivision is pushed off its edge by a hashed offset proportional to the edge
length, which gives a bumpy, non-planar surface with triangles spread over all orientations.  The result depends only on the
input file and `levels`: no library random generator is involved.
"""
from __future__ import annotations

import numpy as np

_U = np.uint64


def read_obj_triangles(path):
    """Minimal reader for triangle-only OBJ files with positive `v` indices (the bunny asset)."""
    verts, faces = [], []
    with open(path, "r") as fh:
        for line in fh:
            if line.startswith("v "):
                t = line.split()
                verts.append((float(t[1]), float(t[2]), float(t[3])))
            elif line.startswith("f "):
                t = line.split()
                faces.append([int(x.split("/")[0]) - 1 for x in t[1:4]])
    return np.array(verts, dtype=np.float64), np.array(faces, dtype=np.int64)


def _hash01(keys, salt):
    """splitmix64 finaliser -> uniform [0,1) doubles."""
    with np.errstate(over="ignore"):
        z = keys.astype(_U) + _U(0x9E3779B97F4A7C15) * _U(salt + 1)
        z = (z ^ (z >> _U(30))) * _U(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U(27))) * _U(0x94D049BB133111EB)
        z = z ^ (z >> _U(31))
    return (z >> _U(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def subdivide(verts, faces, levels, roughness=0.12):
    for level in range(levels):
        nv = verts.shape[0]
        a, b, c = faces[:, 0], faces[:, 1], faces[:, 2]
        e = np.stack([np.stack([a, b], 1), np.stack([b, c], 1), np.stack([c, a], 1)], 1).reshape(-1, 2)
        lo = np.minimum(e[:, 0], e[:, 1])
        hi = np.maximum(e[:, 0], e[:, 1])
        keys = lo * nv + hi
        uniq, inv = np.unique(keys, return_inverse=True)
        ulo, uhi = uniq // nv, uniq % nv
        p0, p1 = verts[ulo], verts[uhi]
        mid = 0.5 * (p0 + p1)
        length = np.sqrt(((p1 - p0) ** 2).sum(1))
        hkeys = uniq * 8 + level
        off = np.stack([_hash01(hkeys, 0), _hash01(hkeys, 1), _hash01(hkeys, 2)], 1) * 2.0 - 1.0
        mid = mid + off * (roughness * length)[:, None]
        m = (nv + inv).reshape(-1, 3)
        mab, mbc, mca = m[:, 0], m[:, 1], m[:, 2]
        faces = np.concatenate([
            np.stack([a, mab, mca], 1),
            np.stack([mab, b, mbc], 1),
            np.stack([mca, mbc, c], 1),
            np.stack([mab, mbc, mca], 1),
        ], 0)
        verts = np.concatenate([verts, mid], 0)
    return verts, faces


STANDIN_EXTENT = 140.0


def dragon_standin(bunny_path, levels=5, extent=STANDIN_EXTENT):
    """Returns (positions float32 [n,3] in dragon.obj units, indices uint32 [m,3])."""
    verts, faces = read_obj_triangles(bunny_path)
    lo, hi = verts.min(0), verts.max(0)
    verts = (verts - 0.5 * (lo + hi)) * (extent / (hi - lo)[1])
    verts, faces = subdivide(verts, faces, levels)
    return verts.astype(np.float32), faces.astype(np.uint32)
