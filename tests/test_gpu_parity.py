"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same inputs.

Bars: primitive/geometry/material ids bit-exact; t, position, normal, uv within 1e-5 relative (they are in fact
expected bit-equal, and the strict tests say so); per-pixel accumulators bit-equal to the oracle's front-to-back
estimator at sizes the oracle finishes in seconds; at full size, properties that do not need the oracle.
"""
import numpy as np
import pytest

from tests.helpers import bits, full_frame_xy, same_bits

pytestmark = pytest.mark.gpu

HIT_FLOAT_FIELDS = ["t", "pos", "n", "uv"]
HIT_ID_FIELDS = ["geom_id", "prim_id", "mat_id"]


def build_pair(oracle, scene_cls, device, **kw):
    from raytracing_amd import api
    so = oracle.OracleScene()
    scene_cls.insert_into(so, **kw)
    so.commit()
    sg = api.HipScene(device)
    scene_cls.insert_into(sg, **kw)
    world = sg.commit().into_world()
    return so, world


def primary_rays(oracle, so, width, height, jitter_sample=None):
    """Un-jittered pixel-centre rays unless a sample index is given."""
    from raytracing_amd import types
    cam = types.Camera.from_dimensions(width, height)
    xy = full_frame_xy(width, height)
    if jitter_sample is None:
        # pixel centres through the same camera arithmetic: dir = normalize((-vx*hw, -vy*hh, -f))
        f = np.float32
        cx = xy[:, 0].astype(np.float32) + f(0.5)
        cy = xy[:, 1].astype(np.float32) + f(0.5)
        vx = f(2.0) * (cx / f(width)) - f(1.0)
        vy = f(2.0) * (cy / f(height)) - f(1.0)
        d = np.stack([-(vx * f(cam.viewport_half_width)), -(vy * f(cam.viewport_half_height)),
                      np.full_like(vx, -cam.focal_length)], 1).astype(np.float32)
        d = d / np.sqrt((d * d).sum(1, dtype=np.float32))[:, None]
        rays = np.zeros(xy.shape[0], dtype=oracle.RAY_DTYPE)
        rays["dir"] = d.astype(np.float32)
    else:
        rays = so.camera_rays(cam, oracle.make_params(spp=4), xy, jitter_sample)
    rays["tnear"] = 1e-5
    rays["tfar"] = np.inf
    return rays


def check_hits(got, want, strict=True):
    for f in HIT_ID_FIELDS:
        assert np.array_equal(got[f], want[f]), "%s differs on %d rays" % (f, (got[f] != want[f]).sum())
    hit = want["geom_id"] != 0xFFFFFFFF
    for f in HIT_FLOAT_FIELDS:
        a, b = got[f][hit].astype(np.float64), want[f][hit].astype(np.float64)
        scale = np.maximum(np.abs(b), 1e-3)
        assert (np.abs(a - b) <= 1e-5 * scale).all(), f
        if strict:
            assert same_bits(got[f], want[f]).all(), "%s not bit-equal" % f
    assert np.isinf(got["t"][~hit]).all()


@pytest.mark.parametrize("name,size", [("cornell-box", (1920, 1080)), ("bunny", (640, 360)), ("spheres", (640, 360)),
                                       ("debug", (320, 180))])
def test_primary_hits_match_oracle(oracle, scene_classes, device, name, size):
    so, world = build_pair(oracle, scene_classes[name], device)
    rays = primary_rays(oracle, so, *size)
    want = so.intersect(rays, oracle.ISECT_BVH)
    got = world.intersect(rays)
    check_hits(got, want)
    assert (want["geom_id"] != 0xFFFFFFFF).any()


def test_bunny_full_hd_prim_ids_vs_brute_force(oracle, scene_classes, device):
    """1920x1080 un-jittered primary rays: GPU BVH == oracle BVH everywhere, == brute force on a strided subset."""
    so, world = build_pair(oracle, scene_classes["bunny"], device)
    rays = primary_rays(oracle, so, 1920, 1080)
    got = world.intersect(rays)
    want = so.intersect(rays, oracle.ISECT_BVH)
    check_hits(got, want)
    sub = slice(0, None, 23)
    brute = so.intersect(rays[sub], oracle.ISECT_BRUTE)
    check_hits(got[sub], brute)


def test_random_rays_inside_cornell(oracle, scene_classes, device):
    so, world = build_pair(oracle, scene_classes["cornell-box"], device)
    rng = np.random.default_rng(7)
    n = 200000
    rays = np.zeros(n, dtype=oracle.RAY_DTYPE)
    rays["org"] = rng.uniform(-0.45, 0.45, (n, 3)).astype(np.float32) + np.array([0, 0, -0.5], np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    rays["dir"] = d / np.sqrt((d * d).sum(1, dtype=np.float32))[:, None]
    rays["tnear"] = 1e-5
    rays["tfar"] = np.inf
    rays["tfar"][::3] = 0.4  # bounded rays
    check_hits(world.intersect(rays), so.intersect(rays, oracle.ISECT_BRUTE))


def test_axis_aligned_and_degenerate_rays(oracle, scene_classes, device):
    """Directions with zero components, rays starting on surfaces, zero-length range."""
    so, world = build_pair(oracle, scene_classes["cornell-box"], device)
    dirs = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [1, 1, 0], [0, -1, -1]],
                    dtype=np.float32)
    dirs = dirs / np.sqrt((dirs * dirs).sum(1))[:, None]
    orgs = np.array([[0, 0, -0.5], [0.1, -0.2, -0.7], [0, -0.5, -0.5], [0.5, 0, -0.5]], dtype=np.float32)
    rays = np.zeros(len(dirs) * len(orgs), dtype=oracle.RAY_DTYPE)
    rays["org"] = np.repeat(orgs, len(dirs), 0)
    rays["dir"] = np.tile(dirs, (len(orgs), 1))
    rays["tnear"] = 1e-5
    rays["tfar"] = np.inf
    rays["tfar"][-3:] = 1e-5
    check_hits(world.intersect(rays), so.intersect(rays, oracle.ISECT_BRUTE))


def render_both(oracle, so, world, device, width, height, spp, max_depth=64, seed=0, integrator=0, xy=None,
                sample_range=None, samples_per_pass=0, cam=None):
    from raytracing_amd import api, types
    cam = cam or types.Camera.from_dimensions(width, height)
    if xy is None:
        xy = full_frame_xy(width, height)
    rng = sample_range or range(0, spp)
    p = oracle.make_params(spp=spp, sample_begin=rng.start, sample_end=rng.stop, max_depth=max_depth, seed=seed,
                           integrator=integrator)
    want, want_rays = so.render(cam, p, xy)
    film = api.Film(device, xy)
    st = api.render(world, film, cam, spp, rng, max_depth, integrator, seed, samples_per_pass)
    got = film.download()
    film.close()
    return got, st, want, want_rays


@pytest.mark.parametrize("name,size,spp,kw", [
    ("cornell-box", (160, 90), 5, {}),
    ("bunny", (160, 90), 20, {}),
    ("spheres", (160, 90), 9, {}),
    ("debug", (96, 54), 4, {}),
    ("dragon", (128, 72), 6, {"levels": 2}),
    ("dragon", (128, 72), 4, {"levels": 2, "diffuse": True}),
])
@pytest.mark.parametrize("tail", ["kernels per bounce", "tail kernel", "tail kernel late"])
def test_accumulators_bit_equal_to_oracle(oracle, scene_classes, device, monkeypatch, name, size, spp, kw, tail):
    """Frames this small fall below the library's tail threshold after the first bounce, so by default every later
    bounce runs in tail_kernel; RTMI_TAIL_RAYS=0 keeps the per-bounce extend / shade kernels to the end, and a
    threshold of a few thousand rays switches in the middle of the path."""
    if tail == "kernels per bounce":
        monkeypatch.setenv("RTMI_TAIL_RAYS", "0")
    elif tail == "tail kernel late":
        monkeypatch.setenv("RTMI_TAIL_RAYS", "3000")
    so, world = build_pair(oracle, scene_classes[name], device, **kw)
    got, st, want, want_rays = render_both(oracle, so, world, device, size[0], size[1], spp)
    assert st["rays"] == want_rays
    assert st["paths"] == size[0] * size[1] * spp
    if tail == "kernels per bounce":
        assert st["tail_rays"] == 0
    elif tail == "tail kernel":
        assert st["tail_rays"] > 0 or st["rays"] == st["paths"]
    eq = same_bits(got, want)
    assert eq.all(), "%d of %d accumulator words differ" % ((~eq).sum(), eq.size)


@pytest.mark.parametrize("name,size,spp,kw", [("cornell-box", (160, 90), 5, {}), ("dragon", (128, 72), 6, {"levels": 2}),
                                              ("spheres", (160, 90), 9, {})])
@pytest.mark.parametrize("queue_depths", ["0", "99"])
def test_both_traversal_kernels_bit_equal_to_oracle(oracle, scene_classes, device, monkeypatch, name, size, spp, kw,
                                                    queue_depths):
    """The library traces the first bounces with the kernel that defers record tests into a per-wave queue and the rest
    with the kernel that votes on its phase; RTMI_QUEUE_DEPTHS moves the switch.  Here every bounce runs in one of them
    (no tail kernel), and both must give the oracle's accumulators."""
    monkeypatch.setenv("RTMI_TAIL_RAYS", "0")
    monkeypatch.setenv("RTMI_QUEUE_DEPTHS", queue_depths)
    so, world = build_pair(oracle, scene_classes[name], device, **kw)
    got, st, want, want_rays = render_both(oracle, so, world, device, size[0], size[1], spp)
    assert st["rays"] == want_rays and st["tail_rays"] == 0
    eq = same_bits(got, want)
    assert eq.all(), "%d of %d accumulator words differ" % ((~eq).sum(), eq.size)


@pytest.mark.parametrize("tail", ["kernels per bounce", "tail kernel", "tail kernel late"])
@pytest.mark.parametrize("aperture", [0.0, 0.1])
def test_thin_and_smooth_dielectrics_and_thin_lens_camera(oracle, device, monkeypatch, tail, aperture):
    """ThinDielectricBxDF, the specular branch and the ior == 1 shortcut of DielectricBxDF (material/mod.rs:249-271,
    339-366), seen through a rotated, off-axis camera with and without a finite aperture (camera.rs:62-85,
    quaternion.rs:12-27): accumulators bit-equal to the oracle, and the camera's rays equal the oracle's."""
    from tests.helpers import OpticsLabScene, lab_camera
    if tail == "kernels per bounce":
        monkeypatch.setenv("RTMI_TAIL_RAYS", "0")
    elif tail == "tail kernel late":
        monkeypatch.setenv("RTMI_TAIL_RAYS", "3000")
    so, world = build_pair(oracle, OpticsLabScene, device)
    W, H, spp = 128, 72, 6
    cam = lab_camera(W, H, aperture)
    got, st, want, want_rays = render_both(oracle, so, world, device, W, H, spp, cam=cam, max_depth=24)
    assert st["rays"] == want_rays and st["rays"] > 2 * W * H * spp
    eq = same_bits(got, want)
    assert eq.all(), "%d of %d accumulator words differ" % ((~eq).sum(), eq.size)
    # every material was reached by a primary ray, and the lens moves the rays when it is open
    xy = full_frame_xy(W, H)
    rays = so.camera_rays(cam, oracle.make_params(spp=spp), xy[::7], 1)
    rays["tnear"] = 1e-5
    hits = world.intersect(rays)
    check_hits(hits, so.intersect(rays, oracle.ISECT_BRUTE))
    assert set(np.unique(hits["mat_id"][hits["geom_id"] != 0xFFFFFFFF])) >= {1, 2, 3, 4, 5}
    pinhole = so.camera_rays(lab_camera(W, H, 0.0), oracle.make_params(spp=spp), xy[::7], 1)
    assert (np.abs(rays["dir"] - pinhole["dir"]).max() > 1e-3) == (aperture > 0)


def test_passes_do_not_change_the_result(oracle, scene_classes, device):
    """One sample per pass, three per pass and everything in one pass give the same accumulators."""
    so, world = build_pair(oracle, scene_classes["cornell-box"], device)
    ref = None
    for spp_pass in (1, 3, 0):
        got, st, want, _ = render_both(oracle, so, world, device, 96, 54, 7, samples_per_pass=spp_pass)
        assert same_bits(got, want).all()
        ref = got if ref is None else ref
        assert same_bits(got, ref).all()


def test_depth_limits_seed_and_range(oracle, scene_classes, device):
    so, world = build_pair(oracle, scene_classes["cornell-box"], device)
    for max_depth in (1, 2, 5):
        got, st, want, want_rays = render_both(oracle, so, world, device, 80, 45, 4, max_depth=max_depth, seed=99)
        assert st["rays"] == want_rays
        assert same_bits(got, want).all()
    # a pixel window that is not tile aligned, and a sample range that does not start at 0
    xy = full_frame_xy(80, 45)
    keep = (xy[:, 0] >= 13) & (xy[:, 0] < 41) & (xy[:, 1] >= 7) & (xy[:, 1] < 30)
    got, st, want, _ = render_both(oracle, so, world, device, 80, 45, 16, xy=xy[keep], sample_range=range(5, 11))
    assert same_bits(got, want).all()


@pytest.mark.parametrize("sampler,pixel_filter,flags", [("uniform", "box", 0x01), ("pixel-centre", "none", 0x22),
                                                        ("stratified", "triangle", 0x10), ("uniform", "triangle", 0x11)])
def test_sampler_and_filter_kinds(oracle, scene_classes, device, sampler, pixel_filter, flags):
    """The samplers and filters the executor never builds (sampler.rs:18-64, filter.rs:11-55): bit-equal to the oracle."""
    from raytracing_amd import api, types
    so, world = build_pair(oracle, scene_classes["cornell-box"], device)
    W, H, spp = 96, 54, 6
    cam = types.Camera.from_dimensions(W, H)
    xy = full_frame_xy(W, H)
    want, want_rays = so.render(cam, oracle.make_params(spp=spp, flags=flags), xy)
    film = api.Film(device, xy)
    st = api.render(world, film, cam, spp, sampler=sampler, pixel_filter=pixel_filter)
    assert st["rays"] == want_rays
    assert same_bits(film.download(), want).all()
    film.close()


def test_random_walk_integrator(oracle, scene_classes, device):
    """`-i basic` (randomwalk.rs:25-68): bit-equal to the oracle's front-to-back form of the walk."""
    so, world = build_pair(oracle, scene_classes["cornell-box"], device)
    got, st, want, want_rays = render_both(oracle, so, world, device, 96, 54, 4, max_depth=8, integrator=1)
    assert st["rays"] == want_rays
    assert same_bits(got, want).all()


def test_resolve_matches_oracle(oracle, scene_classes, device):
    from raytracing_amd import api, types
    so, world = build_pair(oracle, scene_classes["cornell-box"], device)
    cam = types.Camera.from_dimensions(96, 54)
    xy = full_frame_xy(96, 54)
    film = api.Film(device, xy)
    api.render(world, film, cam, 5)
    acc = film.download()
    lin = film.resolve(srgb=False)
    srgb = film.resolve(srgb=True)
    film.close()
    want_lin = oracle.resolve(acc, srgb=False)
    assert same_bits(lin, want_lin).all()
    np.testing.assert_allclose(srgb, oracle.resolve(acc, srgb=True), rtol=0, atol=2e-6)  # powf differs in the last ulp
    # fewer than 3 samples: Variance is +inf (stat.rs:46-48)
    film = api.Film(device, xy[:64])
    api.render(world, film, cam, 2)
    assert np.isinf(film.resolve()[12]).all()
    film.close()


def test_empty_inputs(oracle, scene_classes, device):
    from raytracing_amd import api, types
    # empty scene: every ray misses, every sample is sky
    sg = api.HipScene(device)
    world = sg.commit().into_world()
    rays = np.zeros(5, dtype=oracle.RAY_DTYPE)
    rays["dir"] = (0, 0, -1)
    rays["tfar"] = np.inf
    hits = world.intersect(rays)
    assert (hits["geom_id"] == 0xFFFFFFFF).all() and np.isinf(hits["t"]).all()
    cam = types.Camera.from_dimensions(32, 18)
    film = api.Film(device, full_frame_xy(32, 18))
    st = api.render(world, film, cam, 3)
    acc = film.download()
    assert st["rays"] == 32 * 18 * 3
    np.testing.assert_allclose(acc[6:9].T, np.tile([1.5, 0.9, 3.0], (32 * 18, 1)), rtol=1e-6)
    # zero samples, zero pixels, zero rays
    assert api.render(world, film, cam, 3, range(2, 2))["rays"] == 0
    film.close()
    empty = api.Film(device, np.zeros((0, 2), np.uint32))
    assert api.render(world, empty, cam, 3)["rays"] == 0
    assert empty.download().shape == (22, 0)
    empty.close()
    assert world.intersect(np.zeros(0, dtype=oracle.RAY_DTYPE)).shape == (0,)


def test_unknown_sampler_kind_is_an_argument_error(device, scene_classes):
    from raytracing_amd import _lib, api, types
    import ctypes as C
    sg = api.HipScene(device)
    scene_classes["spheres"].insert_into(sg)
    world = sg.commit().into_world()
    film = api.Film(device, full_frame_xy(8, 8))
    p = _lib.RenderParams()
    p.spp, p.sample_begin, p.sample_end, p.max_depth, p.flags = 1, 0, 1, 4, 0x03
    st = _lib.RenderStats()
    cam = api.camera_pod(types.Camera.from_dimensions(8, 8))
    assert world._l.rtmi_render(world._h, film._h, C.byref(cam), C.byref(p), C.byref(st)) == -1  # RTMI_ERR_ARG
    film.close()


def test_error_behaviour(device):
    from raytracing_amd import api, types
    sg = api.HipScene(device)
    with pytest.raises(api.RtmiError):
        sg.insert_sphere(42, (0, 0, 0), 1.0)  # unknown material
    with pytest.raises(api.RtmiError):
        sg.insert_mesh(0, np.zeros((3, 3), np.float32), np.array([[0, 1, 3]], np.uint32))  # index out of range
    with pytest.raises(api.RtmiError):
        sg.load_obj("/nonexistent.obj", types.Transform(), 0)
    world = sg.commit()
    with pytest.raises(api.RtmiError):
        sg.insert_sphere(0, (0, 0, 0), 1.0)  # immutable once committed
    with pytest.raises(api.RtmiError):
        sg.commit()
    # coordinates the tree's boxes cannot hold: refused at commit, not culled wrongly
    for bad in (np.inf, 3e14):
        sg4 = api.HipScene(device)
        v = np.array([[0, 0, 0], [1, 0, 0], [0, bad, 0]], np.float32)
        sg4.insert_mesh(0, v, np.array([[0, 1, 2]], np.uint32))
        with pytest.raises(api.RtmiError) as err:
            sg4.commit()
        assert err.value.code == -6  # RTMI_ERR_LIMIT
    # a NaN corner makes its triangle unhittable and nothing else
    sg5 = api.HipScene(device)
    v = np.array([[0, 0, -3], [1, 0, -3], [0, 1, -3], [np.nan, 0, -2], [1, 0, -2], [0, 1, -2]], np.float32)
    sg5.insert_mesh(0, v, np.array([[0, 1, 2], [3, 4, 5]], np.uint32))
    w5 = sg5.commit().into_world()
    hit = w5.intersection_full(((0.2, 0.2, 0.0), (0.0, 0.0, -1.0)))
    assert hit is not None and hit["prim_id"] == 0 and hit["t"] == 3.0
    # a progress callback that cancels
    sg2 = api.HipScene(device)
    sg2.insert_sphere(0, (0, 0, 0), 1.0)
    with pytest.raises(api.RtmiError):
        sg2.commit_with_progress(lambda amount: False)
    seen = []
    sg3 = api.HipScene(device)
    sg3.insert_sphere(0, (0, 0, 0), 1.0)
    sg3.commit_with_progress(lambda amount: seen.append(amount) or True)
    assert seen and seen[-1] == 1.0 and all(0.0 <= a <= 1.0 for a in seen)


def test_full_size_properties(oracle, scene_classes, device):
    """Bunny 1920x1080 at 20 spp (one of the benchmark configurations): properties that hold without the oracle.
    - determinism: two renders are bit-identical;
    - RaySeries::merge: samples 0..8 plus samples 8..20 equal 0..20 in every plane that is a plain sum;
    - tiles: rendering the interleaved halves of the tile list separately reproduces the whole frame bit for bit;
    - every pixel accumulated exactly spp samples; the oracle agrees bit for bit on a strided subset of pixels."""
    from raytracing_amd import api, types
    so, world = build_pair(oracle, scene_classes["bunny"], device)
    W, H, spp = 1920, 1080, 20
    cam = types.Camera.from_dimensions(W, H)
    ex = api.Executor((W, H), spp=spp, camera=cam)
    st = ex.run_multithreaded(world, None, batch_size=0)
    whole, xy = ex.last_accum, ex.last_xy
    assert (whole[21] == spp).all() and (whole[9] == spp).all()
    st2 = ex.run_multithreaded(world, None, batch_size=0)
    assert st2["rays"] == st["rays"]
    assert same_bits(ex.last_accum, whole).all()
    # sample-range merge
    film = api.Film(device, xy)
    api.render(world, film, cam, spp, range(0, 8))
    api.render(world, film, cam, spp, range(8, 20))
    assert same_bits(film.download(), whole).all()
    film.close()
    # tile partition
    tiler = api.Tiler(0, 0, W, H, 32, 32)
    n_tiles = tiler.tile_count()
    assert n_tiles == 60 * 34
    out = np.zeros_like(whole)
    index = {(int(x), int(y)): i for i, (x, y) in enumerate(xy)} if False else None
    lin_whole = xy[:, 1].astype(np.int64) * W + xy[:, 0]
    order_whole = np.argsort(lin_whole)
    rays = 0
    parts = []
    for r in range(2):
        exr = api.Executor((W, H), spp=spp, camera=cam)
        s = exr.run_multithreaded(world, None, tiles=range(r, n_tiles, 2), batch_size=0)
        rays += s["rays"]
        parts.append((exr.last_xy, exr.last_accum))
    assert rays == st["rays"]
    pxy = np.concatenate([p[0] for p in parts], 0)
    pacc = np.concatenate([p[1] for p in parts], 1)
    order_parts = np.argsort(pxy[:, 1].astype(np.int64) * W + pxy[:, 0])
    assert same_bits(pacc[:, order_parts], whole[:, order_whole]).all()
    # oracle on a strided subset
    sub = np.arange(0, xy.shape[0], 997)
    want, _ = so.render(cam, oracle.make_params(spp=spp), xy[sub])
    assert same_bits(whole[:, sub], want).all()


def test_triangle_pairing_keeps_every_triangle_exact(oracle, device):
    """The GPU's records pair triangles that share an edge; rotated, reversed, duplicate, non-manifold and degenerate
    triangles must still give the brute-force oracle's hits bit for bit, and a render must give its accumulators."""
    from raytracing_amd import api, types
    from tests.helpers import insert_pairing_torture, torture_rays
    so, sg = oracle.OracleScene(), api.HipScene(device)
    for s in (so, sg):
        m = s.insert_material(types.MaterialDescriptor("m", types.DiffuseBxDF((0.7, 0.6, 0.5))))
        insert_pairing_torture(s, m)
    so.commit()
    world = sg.commit().into_world()
    rays = torture_rays(oracle.RAY_DTYPE)
    got = world.intersect(rays)
    want = so.intersect(rays, oracle.ISECT_BRUTE)
    assert got.tobytes() == want.tobytes()
    W, H = 64, 36
    cam = types.Camera.from_dimensions(W, H)
    xy = full_frame_xy(W, H)
    acc_o, rays_o = so.render(cam, oracle.make_params(spp=4), xy)
    film = api.Film(device, xy)
    st = api.render(world, film, cam, 4)
    assert st["rays"] == rays_o
    assert same_bits(film.download(), acc_o).all()


@pytest.mark.parametrize("mesh,gain", [("mixed", None), ("mixed", "1.3"), ("needles", "1.3")])
def test_spatial_splits_keep_every_hit(oracle, device, monkeypatch, mesh, gain):
    """Trees whose builder cut records at planes (spatial splits; a cut record is held once per part, and a ray may
    test it more than once): hits equal brute force bit for bit, a render equals the oracle's accumulators and ray
    count.  Default settings cut the mixed-scale mesh because it pays; RTMI_SPATIAL_GAIN above 1 forces many more cuts."""
    from raytracing_amd import api, types
    from tests.helpers import cube_rays, mixed_scale_mesh, needle_mesh
    if gain is not None:
        monkeypatch.setenv("RTMI_SPATIAL_GAIN", gain)
    v, t = mixed_scale_mesh() if mesh == "mixed" else needle_mesh()
    so, sg = oracle.OracleScene(), api.HipScene(device)
    for s in (so, sg):
        m = s.insert_material(types.MaterialDescriptor("m", types.DiffuseBxDF((0.7, 0.6, 0.5))))
        s.insert_mesh(m, v, t)
        s.insert_sphere(m, (1.0, -2.0, 0.5), 1.5)
    so.commit()
    world = sg.commit().into_world()
    info = world.info()
    assert info["spatial_splits"] > (10 if gain is None else 100) and info["n_records"] > t.shape[0] + 1
    rays = cube_rays(oracle.RAY_DTYPE)
    want = so.intersect(rays, oracle.ISECT_BRUTE)
    assert (want["geom_id"] != 0xFFFFFFFF).mean() > 0.05
    assert world.intersect(rays).tobytes() == want.tobytes()
    W, H = 64, 36
    cam = types.Camera.from_dimensions(W, H)
    xy = full_frame_xy(W, H)
    acc_o, rays_o = so.render(cam, oracle.make_params(spp=4), xy)
    film = api.Film(device, xy)
    st = api.render(world, film, cam, 4)
    assert st["rays"] == rays_o
    assert same_bits(film.download(), acc_o).all()
    monkeypatch.setenv("RTMI_SPATIAL", "0")
    plain = api.HipScene(device)
    m = plain.insert_material(types.MaterialDescriptor("m", types.DiffuseBxDF((0.7, 0.6, 0.5))))
    plain.insert_mesh(m, v, t)
    plain.insert_sphere(m, (1.0, -2.0, 0.5), 1.5)
    info = plain.commit().into_world().info()
    assert info["spatial_splits"] == 0 and info["n_records"] == t.shape[0] + 1


@pytest.mark.parametrize("n_tri,n_sph", [(1, 0), (0, 1), (2, 0), (1, 1), (3, 0), (2, 2), (5, 1), (9, 3)])
def test_tiny_scenes_match_brute_force(oracle, device, n_tri, n_sph):
    """Trees whose root has one, two, three or four children (and the first two-level trees): the packed node names its
    children by count and first item, so the small counts get a test of their own.  Hits equal brute force, bit for
    bit, and a small render equals the oracle's."""
    from raytracing_amd import api, types
    from tests.helpers import insert_tiny_scene, tiny_scene_rays
    so, sg = oracle.OracleScene(), api.HipScene(device)
    for s in (so, sg):
        m = s.insert_material(types.MaterialDescriptor("m", types.DiffuseBxDF((0.7, 0.6, 0.5))))
        insert_tiny_scene(s, m, n_tri, n_sph)
    so.commit()
    world = sg.commit().into_world()
    rays = tiny_scene_rays(oracle.RAY_DTYPE)
    want = so.intersect(rays, oracle.ISECT_BRUTE)
    assert world.intersect(rays).tobytes() == want.tobytes()
    W, H = 64, 36
    cam = types.Camera.from_dimensions(W, H)
    xy = full_frame_xy(W, H)
    acc_o, rays_o = so.render(cam, oracle.make_params(spp=4), xy)
    film = api.Film(device, xy)
    st = api.render(world, film, cam, 4)
    assert st["rays"] == rays_o
    assert same_bits(film.download(), acc_o).all()


def test_deep_traversal_stack_spills(oracle, device):
    """60,000 large overlapping triangles: rays through the soup need more stack entries than the kernels keep in LDS
    (24 per lane), so the deeper entries go through the global spill buffer.  Hits must still equal brute force."""
    from raytracing_amd import api
    rng = np.random.default_rng(5)
    n = 60000
    c = rng.uniform(-1, 1, (n, 3))
    v = (c[:, None, :] + rng.normal(scale=1.5, size=(n, 3, 3))).astype(np.float32)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    so = oracle.OracleScene()
    so.insert_mesh(0, v.reshape(-1, 3), idx)
    so.commit()
    sg = api.HipScene(device)
    sg.insert_mesh(0, v.reshape(-1, 3), idx)
    world = sg.commit().into_world()
    assert world.info()["stack_need"] > 24
    m = 4096
    rays = np.zeros(m, dtype=oracle.RAY_DTYPE)
    d = rng.normal(size=(m, 3))
    rays["dir"] = d / np.linalg.norm(d, axis=1)[:, None]
    rays["org"] = rng.uniform(-1, 1, (m, 3)) - 3 * rays["dir"]
    rays["tnear"] = 1e-5
    rays["tfar"] = np.inf
    check_hits(world.intersect(rays), so.intersect(rays, oracle.ISECT_BRUTE))


@pytest.mark.parametrize("tail_rays,queue_depths", [("0", "0"), ("0", "99"), ("1000000", "2")])
def test_deep_traversal_stack_in_render(oracle, device, monkeypatch, tail_rays, queue_depths):
    """The same kind of soup, diffuse, rendered from inside: the spill path of both per-bounce traversal kernels
    (RTMI_TAIL_RAYS=0; the one that queues its record tests keeps fewer stack entries in LDS) and of the tail kernel,
    against the oracle's accumulators."""
    from raytracing_amd import api, types
    monkeypatch.setenv("RTMI_TAIL_RAYS", tail_rays)
    monkeypatch.setenv("RTMI_QUEUE_DEPTHS", queue_depths)
    rng = np.random.default_rng(6)
    n = 30000
    c = rng.uniform(-1, 1, (n, 3)) + np.array([0, 0, -2.5])
    v = (c[:, None, :] + rng.normal(scale=1.5, size=(n, 3, 3))).astype(np.float32)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    so, sg = oracle.OracleScene(), api.HipScene(device)
    for s in (so, sg):
        m = s.insert_material(types.MaterialDescriptor("grey", types.DiffuseBxDF((0.8, 0.8, 0.8))))
        s.insert_mesh(m, v.reshape(-1, 3), idx)
    so.commit()
    world = sg.commit().into_world()
    assert world.info()["stack_need"] > 24
    W, H, spp = 48, 27, 2
    cam = types.Camera.from_dimensions(W, H)
    xy = full_frame_xy(W, H)
    want, want_rays = so.render(cam, oracle.make_params(spp=spp, max_depth=6), xy)
    film = api.Film(device, xy)
    st = api.render(world, film, cam, spp, max_depth=6)
    assert st["rays"] == want_rays and st["rays"] > 3 * W * H * spp
    assert same_bits(film.download(), want).all()
    film.close()


def test_allowed_error_adaptive_sampling(oracle, scene_classes, device):
    """--allowed-error: within a batch a pixel stops accumulating once all three channels are precise
    (executor.rs:244-249); batches of 32 as the reference's SampleCounter.  Sky pixels (zero variance) stop at 16
    samples, then take one sample per later batch."""
    from raytracing_amd import api, types
    so, world = build_pair(oracle, scene_classes["debug"], device)
    W, H, spp = 96, 54, 70
    cam = types.Camera.from_dimensions(W, H)
    xy = full_frame_xy(W, H)
    want = None
    film = api.Film(device, xy)
    for a, b in ((0, 32), (32, 64), (64, 70)):
        p = oracle.make_params(spp=spp, sample_begin=a, sample_end=b, allowed_error=0.05)
        want, _ = so.render(cam, p, xy, want)
        api.render(world, film, cam, spp, range(a, b), allowed_error=0.05)
    got = film.download()
    film.close()
    assert same_bits(got, want).all()
    counts = got[21]
    assert counts.min() == 16 + 1 + 1 and counts.max() > 18  # sky: 16, then one per later batch
    # the executor applies it per batch of 32 by itself
    ex = api.Executor((W, H), spp=spp, camera=cam, allowed_error=0.05)
    ex.run_multithreaded(world, None)
    order = np.argsort(ex.last_xy[:, 1].astype(np.int64) * W + ex.last_xy[:, 0])
    assert same_bits(ex.last_accum[:, order], want).all()


def test_cli_writes_the_reference_outputs(device, tmp_path):
    """`rt-cli --scene debug --spp 4 -d 64x36 -o file`: seven channels, LDR JPEG + HDR EXR files, RGB channels sRGB-encoded
    (output/mod.rs:46-51), sky colour (186, 149, 254) as in the reference's renders."""
    from PIL import Image
    from raytracing_amd import cli
    assert cli.main(["--scene", "debug", "--spp", "4", "-d", "64x36", "-o", "file", "--outdir", str(tmp_path),
                     "--seed", "3", "-r", "0..64x0..36"]) == 0
    names = ["Normal", "Position", "Albedo", "Color", "Variance", "Z", "RayDepth"]
    for n in names:
        assert (tmp_path / "ldr" / (n + ".jpeg")).exists() and (tmp_path / "hdr" / (n + ".exr")).exists()
    color = np.load(tmp_path / "hdr" / "Color.npy")
    from tests.helpers import read_exr
    exr = read_exr(tmp_path / "hdr" / "Color.exr")
    assert all(same_bits(exr[c], color[:, :, k]).all() for k, c in enumerate("RGB"))
    assert color.shape == (36, 64, 3)
    corner = np.round(color[0, 0] * 255)
    assert tuple(corner) == (188.0, 149.0, 255.0) or np.allclose(corner, (188, 149, 255), atol=2)
    jpg = np.asarray(Image.open(tmp_path / "ldr" / "Color.jpeg"))
    assert jpg.shape == (36, 64, 3) and abs(int(jpg[0, 0, 2]) - 255) <= 3
    z = np.load(tmp_path / "hdr" / "Z.npy")
    assert z.shape == (36, 64) and z.max() > 0.5 and z[0, 0] == 0.0


def test_handles_may_be_destroyed_in_any_order(device):
    """A scene and a film outlive the device handle they were created on."""
    from raytracing_amd import api, types
    dev2 = api.Device(0)
    sg = api.HipScene(dev2)
    m = sg.insert_material(types.MaterialDescriptor("m", types.DiffuseBxDF((0.5, 0.5, 0.5))))
    sg.insert_sphere(m, (0.0, 0.0, 3.0), 1.0)
    world = sg.commit().into_world()
    xy = full_frame_xy(16, 9)
    film = api.Film(dev2, xy)
    dev2.close()
    st = api.render(world, film, types.Camera.from_dimensions(16, 9), 2)
    assert st["paths"] == 2 * 16 * 9
    film.close()
    sg.close()


_RCCL_GATHER = r"""
import socket, sys
import torch
import torch.distributed as dist
sys.path.insert(0, %r)
from raytracing_amd import api
from raytracing_amd import distributed as D
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
W, H = 70, 50
tiler = api.Tiler(0, 0, W, H, 32, 32)
xy = D.tile_pixels(tiler, range(tiler.tile_count()), (8, 8))
acc = torch.arange(api.ACCUM_PLANES * xy.shape[0], dtype=torch.float32, device="cuda").reshape(api.ACCUM_PLANES, -1)
want = D.FrameGatherer(xy, W, H, "cuda").gather(acc)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%%d" %% port, rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
g = D.FrameGatherer(xy, W, H, "cuda", always_collective=True)
assert g.distributed
buf = torch.zeros_like(want)
got = g.gather(acc * 2.0, buf)
got = g.gather(acc, buf)
t = torch.tensor([3, 4], dtype=torch.int64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.SUM)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(got, want) and t.tolist() == [3, 4]
dist.destroy_process_group()
print("rccl gather ok")
"""


def test_frame_gather_over_rccl():
    """The collectives bench.py uses for N > 1 (all_gather, gather, all_reduce, barrier) on GPU tensors over the nccl
    (RCCL) backend, in a group of one and in a process of their own: the gathered frame must equal the local scatter."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _RCCL_GATHER % root], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0 and "rccl gather ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
