#!/usr/bin/env python3
"""How much would sorting the rays of a deep bounce buy?  Second-generation rays inside the glass dragon (origins =
hit points of rays that already crossed the body once, directions random) are traced with intersect_kernel in three
orders: as generated (pixel order of their grandparent camera ray), shuffled, and sorted by a Morton code of the origin
cell plus the direction octant.  Prints the time of each."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracing_amd import api, scenes, types  # noqa: E402
from raytracing_amd._lib import HIT_DTYPE, RAY_DTYPE  # noqa: E402


def morton3(c, bits):
    out = np.zeros(c.shape[0], dtype=np.uint64)
    for b in range(bits):
        for a in range(3):
            out |= ((c[:, a].astype(np.uint64) >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return out


def main():
    dev = api.Device(0)
    sg = api.HipScene(dev)
    scenes.DragonScene.insert_into(sg)
    world = sg.commit().into_world()
    W, H = 1920, 1080
    cam = types.Camera.from_dimensions(W, H)
    # camera rays through pixel centres (pinhole: origin = lens, direction through the pixel)
    ys, xs = np.mgrid[0:H, 0:W]
    rng = np.random.default_rng(0)
    film = api.Film(dev, np.stack([xs.ravel(), ys.ravel()], 1).astype(np.uint32))
    api.render(world, film, cam, 1, max_depth=1)
    res = film.resolve()
    pos = res[3:6].T.copy()
    z = res[13]
    seen = np.isfinite(z) & (z > 0)
    lens = np.asarray(cam.center_of_lens, dtype=np.float32)
    probe = np.zeros(int(seen.sum()), dtype=RAY_DTYPE)
    d0 = pos[seen] - lens
    probe["org"] = lens
    probe["dir"] = d0 / np.linalg.norm(d0, axis=1)[:, None]
    probe["tnear"] = 1e-5
    probe["tfar"] = np.inf
    h0 = world.intersect(probe)
    big = h0["prim_id"] > 100000  # only the dragon mesh has that many triangles
    pos, nrm = h0["pos"][big], h0["n"][big]
    print("first hits used:", pos.shape[0])
    reps = max(1, 6_000_000 // pos.shape[0])
    pos = np.repeat(pos, reps, 0)
    nrm = np.repeat(nrm, reps, 0)
    n = pos.shape[0]

    def trace(rays_np, label, repeat=3):
        d_r = torch.from_numpy(rays_np.view(np.float32).reshape(n, 8)).cuda()
        d_h = torch.empty((n, 12), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(repeat):
            t0 = time.perf_counter()
            world.intersect_device(d_r.data_ptr(), d_h.data_ptr(), n)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print("%-34s %7.2f ms  %6.0f Mrays/s" % (label, best * 1e3, n / best / 1e6))
        return d_h.cpu().numpy().view(HIT_DTYPE).reshape(n)

    def make(org, direction):
        r = np.zeros(n, dtype=RAY_DTYPE)
        r["org"] = org
        r["dir"] = direction / np.linalg.norm(direction, axis=1)[:, None]
        r["tnear"] = 1e-5
        r["tfar"] = np.inf
        return r

    gen1 = make(pos, -nrm + 0.6 * rng.normal(size=nrm.shape))
    h1 = trace(gen1, "generation 1 (pixel order)")
    ok = np.isfinite(h1["t"])
    org2 = np.where(ok[:, None], h1["pos"], pos)
    n2 = np.where(ok[:, None], h1["n"], nrm)
    inward = -n2 * np.sign(np.sum(n2 * gen1["dir"], 1))[:, None]
    gen2 = make(org2, inward + 0.6 * rng.normal(size=nrm.shape))
    trace(gen2, "generation 2 (pixel order)")
    perm = rng.permutation(n)
    trace(gen2[perm], "generation 2 shuffled")
    lo, hi = org2.min(0), org2.max(0)
    for bits in (4, 6, 8):
        cell = np.clip(((org2 - lo) / (hi - lo + 1e-9) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
        key = morton3(cell, bits)
        octant = ((gen2["dir"][:, 0] < 0).astype(np.uint64) | ((gen2["dir"][:, 1] < 0).astype(np.uint64) << np.uint64(1)) |
                  ((gen2["dir"][:, 2] < 0).astype(np.uint64) << np.uint64(2)))
        trace(gen2[np.argsort(key, kind="stable")], "sorted by origin cell (%d bits/axis)" % bits)
        trace(gen2[np.argsort((key << np.uint64(3)) | octant, kind="stable")], "  ... then direction octant")
        trace(gen2[np.argsort((octant << np.uint64(3 * bits)) | key, kind="stable")], "  octant first, then cell")


if __name__ == "__main__":
    main()
