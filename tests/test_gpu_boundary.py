"""GPU tests of the boundary's behaviour around the hot path: concurrent device-side queries on deep trees, the
per-ray query seam, the documented ray-origin range, adaptive sampling across passes."""
import ctypes as C
import time

import numpy as np
import pytest

from tests.helpers import full_frame_xy, same_bits

pytestmark = pytest.mark.gpu


def soup(n, seed, offset=(0.0, 0.0, 0.0)):
    """n large overlapping triangles: a tree whose traversal needs more stack than the kernels keep on chip."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-1, 1, (n, 3)) + np.asarray(offset)
    v = (c[:, None, :] + rng.normal(scale=1.5, size=(n, 3, 3))).astype(np.float32)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    return v.reshape(-1, 3), idx


def random_rays(dtype, m, seed):
    rng = np.random.default_rng(seed)
    rays = np.zeros(m, dtype=dtype)
    d = rng.normal(size=(m, 3))
    rays["dir"] = d / np.linalg.norm(d, axis=1)[:, None]
    rays["org"] = rng.uniform(-1, 1, (m, 3)) - 3 * rays["dir"]
    rays["tnear"] = 1e-5
    rays["tfar"] = np.inf
    return rays


def test_overlapping_device_queries_on_a_deep_tree(oracle, device):
    """Six rtmi_intersect_device launches on three streams, none waiting for another on the host, on a tree that
    spills its traversal stack to global memory: every launch must give what the same rays give one batch at a time
    (itself checked against brute force on a subset)."""
    from raytracing_amd import api, _lib
    from tests import hip_rt
    v, idx = soup(60000, 5)
    sg = api.HipScene(device)
    sg.insert_mesh(0, v, idx)
    world = sg.commit().into_world()
    assert world.info()["stack_need"] > 24
    so = oracle.OracleScene()
    so.insert_mesh(0, v, idx)
    so.commit()
    m = 150000
    batches = [random_rays(_lib.RAY_DTYPE, m, 100 + k) for k in range(6)]
    want = [world.intersect(b) for b in batches]
    sub = slice(0, None, 97)
    brute = so.intersect(batches[0][sub], oracle.ISECT_BRUTE)
    assert want[0][sub].tobytes() == brute.tobytes()
    streams = [hip_rt.Stream() for _ in range(3)]
    d_rays = [hip_rt.DeviceBuffer.from_array(b) for b in batches]
    d_hits = [hip_rt.DeviceBuffer(m * _lib.HIT_DTYPE.itemsize) for _ in batches]
    for h in d_hits:
        h.fill(0xFF)
    for k in range(6):
        world.intersect_device(d_rays[k].ptr, d_hits[k].ptr, m, C.c_void_p(streams[k % 3].handle))
    for s in streams:
        s.synchronize()
    for k in range(6):
        got = d_hits[k].download(_lib.HIT_DTYPE)
        assert got.tobytes() == want[k].tobytes(), "launch %d differs" % k
    for b in d_rays + d_hits:
        b.free()
    for s in streams:
        s.destroy()


def test_single_ray_queries(oracle, scene_classes, device):
    """Shape::intersection_full one ray at a time (what a CPU-side integrator of the reference would do): same hits as
    the batch, and cheap enough to use -- the scene keeps its staging buffers between calls."""
    from raytracing_amd import api, _lib
    sg = api.HipScene(device)
    scene_classes["cornell-box"].insert_into(sg)
    world = sg.commit().into_world()
    rays = random_rays(_lib.RAY_DTYPE, 2000, 3)
    rays["org"] = rays["org"] * 0.1 + np.array([0, 0, -0.5], np.float32)
    batch = world.intersect(rays)
    world.intersect(rays[:1])  # creates the staging pair
    t0 = time.perf_counter()
    for i in range(2000):
        one = world.intersect(rays[i:i + 1])
        assert one.tobytes() == batch[i:i + 1].tobytes()
    per_call = (time.perf_counter() - t0) / 2000
    assert per_call < 2e-3, "single-ray query takes %.0f us" % (per_call * 1e6)
    hit = world.intersection_full((rays["org"][0], rays["dir"][0], (1e-5, np.inf)))
    assert (hit is None) == (batch["geom_id"][0] == 0xFFFFFFFF)


def test_ray_origin_range_is_enforced(scene_classes, device):
    """Culling boxes are padded for ray origins within 16 scene extents; rtmi_intersect refuses rays from further out
    instead of risking a culled hit, and accepts the same ray started nearer."""
    from raytracing_amd import api, _lib
    sg = api.HipScene(device)
    scene_classes["debug"].insert_into(sg)
    world = sg.commit().into_world()
    lo, hi = world.bounding_box()
    extent = float(max(np.abs(lo).max(), np.abs(hi).max()))
    rays = np.zeros(2, dtype=_lib.RAY_DTYPE)
    rays["dir"] = (0.0, 0.0, -1.0)
    rays["org"][0] = (0.0, 0.0, 15.0 * extent)
    rays["org"][1] = (0.0, 0.0, 17.0 * extent)
    rays["tfar"] = np.inf
    with pytest.raises(api.RtmiError) as e:
        world.intersect(rays)
    assert e.value.code == -1 and "ray 1" in str(e.value)
    hits = world.intersect(rays[:1])
    assert hits["geom_id"][0] == 0 and abs(hits["t"][0] - (15.0 * extent + 0.7)) < 1e-3


def test_allowed_error_across_passes(oracle, scene_classes, device):
    """--allowed-error when one call is split into passes: a pixel that became precise in an early pass takes no
    further sample in that call (executor.rs:244-249 breaks out of the batch), whatever samples_per_pass is, and the
    passes after it no longer trace the pixel."""
    from raytracing_amd import api, types
    so = oracle.OracleScene()
    scene_classes["debug"].insert_into(so)
    so.commit()
    sg = api.HipScene(device)
    scene_classes["debug"].insert_into(sg)
    world = sg.commit().into_world()
    W, H, spp = 96, 54, 64
    cam = types.Camera.from_dimensions(W, H)
    xy = full_frame_xy(W, H)
    want = None
    want_rays = 0
    for a, b in ((0, 32), (32, 64)):
        want, r = so.render(cam, oracle.make_params(spp=spp, sample_begin=a, sample_end=b, allowed_error=0.05), xy, want)
        want_rays += r
    traced = {}
    for spp_pass in (32, 5, 1, 0):
        film = api.Film(device, xy)
        rays = 0
        for a, b in ((0, 32), (32, 64)):
            st = api.render(world, film, cam, spp, range(a, b), allowed_error=0.05, samples_per_pass=spp_pass)
            rays += st["rays"]
            assert st["passes"] == {5: 7, 1: 32, 0: 4, 32: 1}[spp_pass]
        got = film.download()
        film.close()
        assert same_bits(got, want).all(), "samples_per_pass=%d" % spp_pass
        traced[spp_pass] = rays
    # one pass traces every sample; one sample per pass traces exactly what the reference would; the default (passes of
    # 8) and passes of 5 lie in between
    assert traced[1] == want_rays
    assert traced[1] < traced[5] < traced[32] and traced[1] < traced[0] < traced[32]


def test_limits_are_reported(device, scene_classes):
    from raytracing_amd import api, types
    sg = api.HipScene(device)
    scene_classes["debug"].insert_into(sg)
    world = sg.commit().into_world()
    film = api.Film(device, full_frame_xy(8, 8))
    with pytest.raises(api.RtmiError) as e:
        api.render(world, film, types.Camera.from_dimensions(8, 8), 1, max_depth=4097)
    assert e.value.code == -6
    st = api.render(world, film, types.Camera.from_dimensions(8, 8), 1, max_depth=4096)
    assert st["paths"] == 64
    film.close()


def test_tile_films_and_frame_gather(oracle, scene_classes, device):
    """The multi-GPU pieces of the C ABI on one GPU: the tile share of every rank of a 3-rank job as films, rendered one
    after the other, gathered into one frame (no communicator: films and frame on one device), must equal the
    whole frame rendered at once, bit for bit; and the pixel lists are those of the Python partition."""
    from raytracing_amd import api, types
    from raytracing_amd import distributed as D
    sg = api.HipScene(device)
    scene_classes["cornell-box"].insert_into(sg)
    world = sg.commit().into_world()
    W, H, spp, grain = 150, 85, 3, 32  # ragged right and bottom tiles
    cam = types.Camera.from_dimensions(W, H)
    tiler = api.Tiler(0, 0, W, H, grain, grain)
    n_tiles = tiler.tile_count()
    frame = api.Film.frame(device, W, H)
    assert np.array_equal(frame.xy, full_frame_xy(W, H))
    ranks = 3
    rays = 0
    for r in range(ranks):
        tiles = api.tile_partition(n_tiles, ranks, r)
        assert list(tiles) == list(D.partition_tiles(n_tiles, ranks, r))
        film = api.Film.for_tiles(device, tiler, tiles, (8, 8))
        assert np.array_equal(film.xy, D.tile_pixels(tiler, tiles, (8, 8)))
        rays += api.render(world, film, cam, spp)["rays"]
        film.gather(None, frame)
        film.gather(None, frame)  # the plan is reused
        film.close()
    whole = api.Film(device, full_frame_xy(W, H))
    st = api.render(world, whole, cam, spp)
    assert st["rays"] == rays
    assert same_bits(frame.download(), whole.download()).all()
    # row-major tiles without blocks, a window with an offset, more ranks than tiles
    t2 = api.Tiler(7, 5, 40, 33, 32, 32)
    f2 = api.Film.for_tiles(device, t2, api.tile_partition(t2.tile_count(), 2, 1), None)
    assert np.array_equal(f2.xy, D.tile_pixels(t2, D.partition_tiles(t2.tile_count(), 2, 1)))
    assert len(api.tile_partition(4, 8, 6)) == 0 and api.Film.for_tiles(device, t2, range(6, 4, 8)).n_pixels == 0
    # a film with pixels outside the frame is refused
    small = api.Film.frame(device, 16, 16)
    with pytest.raises(api.RtmiError) as e:
        f2.gather(None, small)
    assert e.value.code == -1
    for f in (frame, whole, f2, small):
        f.close()


def test_frame_gather_over_rccl_through_the_c_abi(scene_classes, device):
    """rtmi_comm + rtmi_film_gather with a communicator: a job of one rank, whose send and receive go through RCCL
    exactly as a peer's would (the root receives its own planes like everybody else's)."""
    from raytracing_amd import api, types
    sg = api.HipScene(device)
    scene_classes["debug"].insert_into(sg)
    world = sg.commit().into_world()
    W, H, spp = 96, 54, 2
    cam = types.Camera.from_dimensions(W, H)
    tiler = api.Tiler(0, 0, W, H, 32, 32)
    comm = api.Comm(device, api.Comm.unique_id(), 0, 1)
    film = api.Film.for_tiles(device, tiler, api.tile_partition(tiler.tile_count(), 1, 0))
    frame = api.Film.frame(device, W, H)
    api.render(world, film, cam, spp)
    film.gather(comm, frame)
    first = frame.download()
    whole = api.Film(device, full_frame_xy(W, H))
    api.render(world, whole, cam, spp)
    assert same_bits(first, whole.download()).all()
    api.render(world, film, cam, spp, range(2, 4))  # a second frame through the same plan
    film.gather(comm, frame)
    api.render(world, whole, cam, spp, range(2, 4))
    assert same_bits(frame.download(), whole.download()).all()
    with pytest.raises(api.RtmiError):
        film.gather(comm, frame, root=1)
    # errors only the root can see go through the ranks' status exchange: a frame the film does not fit in, then a
    # frame it fits in again (no new exchange of pixel lists), then a window with an offset
    small = api.Film.frame(device, 32, 32)
    with pytest.raises(api.RtmiError) as e:
        film.gather(comm, small)
    assert e.value.code == -1 and "outside the frame" in str(e.value)
    with pytest.raises(api.RtmiError) as e:
        film.gather(comm, None)
    assert e.value.code == -1
    wide = api.Film.frame(device, W + 8, H + 4)
    film.gather(comm, wide)
    got = wide.download().reshape(22, H + 4, W + 8)
    assert same_bits(got[:, :H, :W].reshape(22, -1), whole.download()).all() and (got[:, H:, :] == 0).all()
    film.gather(comm, frame)
    assert same_bits(frame.download(), whole.download()).all()
    for f in (film, frame, whole, small, wide):
        f.close()
    comm.close()


def test_c_host_runs(tmp_path, scene_classes, device):
    """examples/abi_smoke.c (plain C99 against include/rtmi.h): scene, single-ray query, tile film, render, gather,
    resolve -- and the same ray and path counts as the Python mirror gets for that scene and camera."""
    import re
    import subprocess
    from raytracing_amd import api, types
    from tests.test_host_logic import build_c_host
    exe, env = build_c_host(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=100)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"rendered (\d+) paths, (\d+) rays", r.stdout)
    sg = api.HipScene(device)
    scene_classes["debug"].insert_into(sg)
    world = sg.commit().into_world()
    film = api.Film(device, full_frame_xy(64, 36))
    st = api.render(world, film, types.Camera.from_dimensions(64, 36), 4)
    film.close()
    assert (int(m.group(1)), int(m.group(2))) == (st["paths"], st["rays"])


def test_renders_and_queries_from_several_threads(oracle, scene_classes, device):
    """Shape is Sync + Send (shape/mod.rs:3) and the reference calls it from every rayon worker: four host threads
    render their own films and query the same committed scene at once; the calls serialise inside the library and
    every thread gets what a lone call gets."""
    import threading
    from raytracing_amd import api, types, _lib
    sg = api.HipScene(device)
    scene_classes["cornell-box"].insert_into(sg)
    world = sg.commit().into_world()
    W, H, spp = 64, 36, 3
    cam = types.Camera.from_dimensions(W, H)
    xy = full_frame_xy(W, H)
    rays = random_rays(_lib.RAY_DTYPE, 5000, 11)
    rays["org"] = rays["org"] * 0.1 + np.array([0, 0, -0.5], np.float32)
    film = api.Film(device, xy)
    api.render(world, film, cam, spp)
    want_acc, want_hits = film.download(), world.intersect(rays)
    film.close()
    out, errors = {}, []

    def work(k):
        try:
            f = api.Film(device, xy)
            for _ in range(3):
                f.clear()
                api.render(world, f, cam, spp)
                hits = world.intersect(rays)
            out[k] = (f.download(), hits)
            f.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(4):
        assert same_bits(out[k][0], want_acc).all() and out[k][1].tobytes() == want_hits.tobytes()
