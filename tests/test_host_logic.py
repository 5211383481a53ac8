"""CPU-side tests of the product's host logic: the C ABI's exported surface, the OBJ/MTL loader against the oracle's
independent restatement, the BVH builder, the device code compiled for the host (tests/hostsim) against the oracle,
and the multi-rank tile partition + gather over gloo."""
import ctypes as C
import os
import re
import socket

import numpy as np
import pytest

from tests.helpers import full_frame_xy, same_bits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.fixture(scope="module")
def hostsim():
    from tests.hostsim import hostsim as HS
    HS.build()
    return HS


# ------------------------------------------------------------------------------------------------ C ABI surface

def header_functions():
    text = open(os.path.join(ROOT, "include", "rtmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtmi_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from raytracing_amd import _lib
    L = _lib.load()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "librtmi.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names, "binding table and header disagree"
    assert b"gfx950" in L.rtmi_version()


def test_structure_layouts_match_the_header():
    from raytracing_amd import _lib
    assert C.sizeof(_lib.Material) == 32
    assert _lib.RAY_DTYPE.itemsize == 32 and _lib.HIT_DTYPE.itemsize == 48
    assert C.sizeof(_lib.Camera) == 52 and C.sizeof(_lib.TransformPOD) == 40
    assert C.sizeof(_lib.RenderParams) == 40 and C.sizeof(_lib.RenderStats) == 48
    assert C.sizeof(_lib.SceneInfo) == 64 and C.sizeof(_lib.BounceProfile) == 32


def test_no_gpu_fails_loudly():
    if has_gpu():
        pytest.skip("a GPU is present")
    from raytracing_amd import api
    with pytest.raises(api.RtmiError) as e:
        api.Device(0)
    assert e.value.code == -2 and "no HIP device" in str(e.value)


def test_product_does_not_reference_the_oracle():
    """The shipped package must not import, load or link the oracle."""
    pkg = os.path.join(ROOT, "raytracing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "rt_oracle" not in text and "librt_oracle" not in text, f
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f


# ------------------------------------------------------------------------------------------------ loader

def build_both(oracle, hostsim, scene_cls, **kw):
    so = oracle.OracleScene()
    scene_cls.insert_into(so, **kw)
    so.commit()
    sh = hostsim.HostSimScene()
    scene_cls.insert_into(sh, **kw)
    sh.commit()
    return so, sh


def test_cornell_obj_models_materials(oracle, hostsim, scene_classes):
    """tobj semantics on obj/cornell_box.obj: 8 models (the two boxes split off by `usemtl`), 36 triangles, materials
    in MTL order, default material unused, sphere last."""
    from oracle import obj_tobj
    from raytracing_amd import scenes
    models, mats = obj_tobj.load_obj(scenes.obj_path("cornell_box.obj"))
    assert [m.name for m in models] == ["floor", "ceiling", "backWall", "rightWall", "leftWall", "leftWall",
                                        "shortBox", "light"]
    assert [len(m.indices) for m in models] == [2, 2, 2, 2, 2, 12, 12, 2]
    assert [m.name for m in mats] == ["leftWall", "rightWall", "floor", "ceiling", "backWall", "shortBox", "tallBox",
                                      "light"]
    assert [m.material_id for m in models] == [2, 3, 4, 1, 0, 5, 6, 7]
    assert mats[0].diffuse == (0.63, 0.065, 0.05) and mats[7].diffuse == (0.78, 0.78, 0.78)  # Ke is ignored
    so, sh = build_both(oracle, hostsim, scene_classes["cornell-box"])
    assert sh.geometry_count() == 9
    # MaterialId: 0 Sky, 1 default, 2.. MTL order, 10 "light!"
    assert [sh.geometry_material(g) for g in range(9)] == [4, 5, 6, 3, 2, 7, 8, 9, 10]
    table = sh.material_table()
    assert len(table) == 11 and table[0] == (0, 0.0, 0.0, 0.0, 0.0, 0.0) and table[10][0] == 3
    assert np.allclose(table[3][1:4], (0.14, 0.45, 0.091)) and np.allclose(table[1][1:4], (5.5, 0.8, 0.9))
    # quad (a, b, c, d) -> (a, b, c), (a, c, d); transform scale 0.5, translate (0, -0.5, -0.5)
    floor = sh.geometry_triangles(0)
    assert np.allclose(floor[0], [-0.505, -0.5, -0.005, 0.5, -0.5, -0.005, 0.5, -0.5, -1.02], atol=1e-6)
    assert np.allclose(floor[1], [-0.505, -0.5, -0.005, 0.5, -0.5, -1.02, -0.495, -0.5, -1.02], atol=1e-6)


@pytest.mark.parametrize("name", ["cornell-box", "bunny"])
def test_loader_matches_oracle_restatement_bitwise(oracle, hostsim, scene_classes, name):
    from oracle import obj_tobj
    from raytracing_amd import scenes
    sh = hostsim.HostSimScene()
    scene_classes[name].insert_into(sh)
    path = scenes.obj_path("cornell_box.obj" if name == "cornell-box" else "standford_bunny.obj")
    models, _ = obj_tobj.load_obj(path)
    tr = {"cornell-box": ((0.0, -0.5, -0.5), (0.5,) * 3), "bunny": ((0.2, -0.3, -0.5), (4.0,) * 3)}[name]
    for g, m in enumerate(models):
        pos = obj_tobj.apply_transform(m.positions, tr[0], tr[1], (0, 0, 0, 1))
        want = pos[m.indices.astype(np.int64)].reshape(-1, 9)
        got = sh.geometry_triangles(g)
        assert got.shape == want.shape and same_bits(got, want).all()


def test_obj_edge_cases(oracle, hostsim, tmp_path):
    """Polygons, negative indices, v/vt/vn corners, tabs, comments, a missing MTL, an unknown usemtl, an empty tail."""
    from raytracing_amd import types
    obj = tmp_path / "t.obj"
    obj.write_text(
        "# comment\nmtllib missing.mtl\nv 0 0 0\nv\t1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 1.5 0\n"
        "usemtl nothing\nf 1/1/1 2/2/2 3/3/3 4/4/4 5/5/5\n"
        "o second\nv 0 0 1\nv 1 0 1\nv 0 1 1\nf -3 -2 -1\nf 6//1 7//1 8//1\n"
        "g tail\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl b\nf 1 2 3\nusemtl a\nf 3 2 1\n"
                                    "usemtl zzz\nf 1 3 2\n")
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 0.1 0.2 0.3\nnewmtl b\nKd 0.4 0.5 0.6 # c\n")
    tr = types.Transform(translation=(1, 2, 3), scale=(2, 2, 2),
                         rot=types.quat_from_axis_angle((0, 1, 0), 0.7))
    for fname, n_geoms, mats in (("t.obj", 3, [7, 7, 7]), ("m.obj", 3, [2, 1, 7])):
        so, sh = oracle.OracleScene(), hostsim.HostSimScene()
        for s in (so, sh):
            for _ in range(6):
                s.insert_material(types.DiffuseBxDF())  # ids 1..6
            default = s.insert_material(types.DiffuseBxDF(albedo=(1, 1, 1)))  # id 7
            assert default == 7
        # the oracle inserted MTL materials after the default, the product too: compare per geometry
        ho = so.load_obj(str(tmp_path / fname), tr, 7)
        sh.load_obj(str(tmp_path / fname), tr, 7)
        assert len(ho) == n_geoms == sh.geometry_count()
        got_mats = [sh.geometry_material(g) for g in range(n_geoms)]
        if fname == "m.obj":
            assert got_mats == [9, 8, 7]  # b, a, unknown -> default
        else:
            assert got_mats == mats
    sh = hostsim.HostSimScene()
    sh.load_obj(str(tmp_path / "t.obj"), types.Transform(), 0)
    assert [len(sh.geometry_triangles(g)) for g in range(3)] == [3, 2, 0]  # fan of a pentagon; trailing empty model
    assert np.array_equal(sh.geometry_triangles(0)[2], [0, 0, 0, 0, 1, 0, 0.5, 1.5, 0])
    assert np.array_equal(sh.geometry_triangles(1)[0], sh.geometry_triangles(1)[1])
    with pytest.raises(IOError):
        sh.load_obj(str(tmp_path / "absent.obj"), types.Transform(), 0)
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nv 1 0 0\nf 1 2\n")
    with pytest.raises(IOError):
        sh.load_obj(str(bad), types.Transform(), 0)


# ------------------------------------------------------------------------------------------------ BVH + device code

@pytest.mark.parametrize("name,kw", [("cornell-box", {}), ("bunny", {}), ("spheres", {}), ("debug", {}),
                                     ("dragon", {"levels": 1})])
def test_device_code_on_host_matches_oracle(oracle, hostsim, scene_classes, name, kw):
    """Closest hits and whole accumulators, bit for bit, from the kernels' per-ray code built for the host."""
    from raytracing_amd import types
    so, sh = build_both(oracle, hostsim, scene_classes[name], **kw)
    assert sh.info["n_prims"] == so.prim_count()
    lo_o, hi_o = so.bounding_box()
    lo_h, hi_h = sh.bounding_box()
    assert np.array_equal(lo_o, lo_h) and np.array_equal(hi_o, hi_h)
    W, H = 96, 54
    cam = types.Camera.from_dimensions(W, H)
    xy = full_frame_xy(W, H)
    rays = so.camera_rays(cam, oracle.make_params(spp=1), xy, 0)
    rays["tnear"] = 1e-5
    want = so.intersect(rays, oracle.ISECT_BRUTE if so.prim_count() < 10000 else oracle.ISECT_BVH)
    got = sh.intersect(rays)
    assert got.tobytes() == want.tobytes()
    spp = 3
    acc_o, rays_o = so.render(cam, oracle.make_params(spp=spp), xy)
    acc_h, rays_h = sh.render(cam, hostsim.params(spp=spp), xy)
    assert rays_o == rays_h
    assert same_bits(acc_o, acc_h).all()
    assert sh.stack_high() <= sh.info["stack_need"] <= 128


@pytest.mark.parametrize("aperture", [0.0, 0.1])
def test_optics_lab_on_host_matches_oracle(oracle, hostsim, aperture):
    """Thin dielectric, smooth dielectric, ior == 1 and the thin-lens, rotated camera: the device code built for the
    host against the oracle, bit for bit; and the scene does reach the branches it is meant to reach."""
    from tests.helpers import OpticsLabScene, lab_camera
    so, sh = build_both(oracle, hostsim, OpticsLabScene)
    W, H, spp = 64, 36, 4
    cam = lab_camera(W, H, aperture)
    xy = full_frame_xy(W, H)
    acc_o, rays_o = so.render(cam, oracle.make_params(spp=spp, max_depth=24), xy)
    acc_h, rays_h = sh.render(cam, hostsim.params(spp=spp, max_depth=24), xy)
    assert rays_o == rays_h and rays_o > 2 * W * H * spp
    assert same_bits(acc_o, acc_h).all()
    rays = so.camera_rays(cam, oracle.make_params(spp=spp), xy, 1)
    rays["tnear"] = 1e-5
    hits = so.intersect(rays, oracle.ISECT_BRUTE)
    assert sh.intersect(rays).tobytes() == hits.tobytes()
    assert set(np.unique(hits["mat_id"][hits["geom_id"] != 0xFFFFFFFF])) >= {1, 2, 3, 4, 5}
    # the recursive form of the reference's integrator agrees with the front-to-back one on this scene too
    acc_r, rays_r = so.render(cam, oracle.make_params(spp=spp, max_depth=24, form=oracle.FORM_RECURSIVE), xy)
    assert rays_r == rays_o
    np.testing.assert_allclose(acc_r[6:9], acc_o[6:9], rtol=2e-4, atol=1e-6)


def test_random_walk_on_host_matches_oracle(oracle, hostsim, scene_classes):
    """`-i basic`: the device code's random walk, built for the host, against the oracle's front-to-back walk."""
    from raytracing_amd import types
    so, sh = build_both(oracle, hostsim, scene_classes["cornell-box"])
    cam = types.Camera.from_dimensions(48, 27)
    xy = full_frame_xy(48, 27)
    acc_o, rays_o = so.render(cam, oracle.make_params(spp=4, max_depth=8, integrator=1), xy)
    acc_h, rays_h = sh.render(cam, hostsim.params(spp=4, max_depth=8, integrator=1), xy)
    assert rays_o == rays_h and same_bits(acc_o, acc_h).all()


@pytest.mark.parametrize("flags", [0x01, 0x02, 0x10, 0x20, 0x11, 0x22])
def test_sampler_and_filter_kinds_on_host_match_oracle(oracle, hostsim, scene_classes, flags):
    so, sh = build_both(oracle, hostsim, scene_classes["cornell-box"])
    W, H = 48, 27
    from raytracing_amd import types
    cam = types.Camera.from_dimensions(W, H)
    xy = full_frame_xy(W, H)
    acc_o, _ = so.render(cam, oracle.make_params(spp=4, flags=flags), xy)
    acc_h, _ = sh.render(cam, hostsim.params(spp=4, flags=flags), xy)
    assert same_bits(acc_o, acc_h).all()
    base, _ = so.render(cam, oracle.make_params(spp=4), xy)
    assert not same_bits(acc_o, base).all()


def test_bvh_finds_every_primitive(oracle, hostsim):
    """Random triangles: a ray aimed at each triangle's centroid from just in front of it must report that triangle
    (or a nearer one the brute-force oracle agrees on)."""
    rng = np.random.default_rng(11)
    n = 3000
    c = rng.uniform(-1, 1, (n, 3))
    v = (c[:, None, :] + rng.normal(scale=0.03, size=(n, 3, 3))).astype(np.float32)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    so, sh = oracle.OracleScene(), hostsim.HostSimScene()
    for s in (so, sh):
        s.insert_mesh(0, v.reshape(-1, 3), idx)
        s.insert_sphere(0, (0.2, 0.1, -0.3), 0.25)
        s.commit()
    cen = v.mean(1)
    nrm = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    rays = np.zeros(n, dtype=oracle.RAY_DTYPE)
    rays["org"] = cen + 0.01 * nrm
    rays["dir"] = -nrm
    rays["tnear"] = 1e-5
    rays["tfar"] = np.inf
    got = sh.intersect(rays)
    want = so.intersect(rays, oracle.ISECT_BRUTE)
    assert got.tobytes() == want.tobytes()
    assert (got["prim_id"] == np.arange(n)).mean() > 0.95
    # long random rays through the soup
    d = rng.normal(size=(n, 3))
    rays["org"] = rng.uniform(-2, 2, (n, 3))
    rays["dir"] = d / np.linalg.norm(d, axis=1)[:, None]
    assert sh.intersect(rays).tobytes() == so.intersect(rays, oracle.ISECT_BRUTE).tobytes()
    assert sh.stack_high() <= sh.info["stack_need"]


@pytest.mark.parametrize("name,kw", [("cornell-box", {}), ("bunny", {}), ("dragon", {"levels": 1}), ("mixed-scale mesh", {})])
def test_node_items_name_their_children_by_one_index(hostsim, scene_classes, name, kw):
    """The packed node the kernels read (48 of the item's 64 bytes, csrc/rt_device.h) against the plain child
    references the builder keeps in the item's last four words: child k is item first + k, leaf bits and child count
    agree, the children fill the first slots, a group never straddles more 128-byte lines than it must, the steps are
    powers of two, every item is the child of exactly one node, and every record item holds a record."""
    sh = hostsim.HostSimScene()
    if name in scene_classes:
        scene_classes[name].insert_into(sh, **kw)
    else:  # a tree with spatial splits: cut records are held once per part
        from tests.helpers import mixed_scale_mesh
        sh.insert_mesh(0, *mixed_scale_mesh())
    sh.commit()
    assert name in scene_classes or sh.info["spatial_splits"] > 10
    items = sh.items()
    n_items = items.shape[0]
    EMPTY, LEAF = 0xFFFFFFFF, 0x80000000
    parents = np.zeros(n_items, dtype=np.int64)
    is_node = np.zeros(n_items, dtype=bool)
    todo, n_nodes, n_records = [0], 0, 0
    is_node[0] = True
    while todo:
        it = todo.pop()
        w = items[it]
        n_nodes += 1
        meta, first = int(w[10]), int(w[11])
        leaf_bits, last = (meta >> 16) & 15, (meta >> 24) & 3
        refs = [int(r) for r in w[12:16]]
        present = [r != EMPTY for r in refs]
        assert present == [k <= last for k in range(4)], "children fill slots 0 .. count - 1"
        assert first % 2 == 0 and first + last < n_items, "a group starts on a line boundary"
        for k in range(last + 1):
            assert (refs[k] & 0x0FFFFFFF) == first + k
            assert bool(refs[k] & LEAF) == bool(leaf_bits >> k & 1)
            assert (refs[k] >> 28) & 7 == 0 or not refs[k] & LEAF, "one record per leaf"
            parents[first + k] += 1
            if refs[k] & LEAF:
                n_records += 1
            else:
                is_node[first + k] = True
                todo.append(first + k)
        assert leaf_bits >> (last + 1) == 0
        for half in (int(w[3]) & 0xFFFF, int(w[3]) >> 16, meta & 0xFFFF):
            step = np.array([half << 16], dtype=np.uint32).view(np.float32)[0]
            assert step > 0 and np.log2(float(step)) == round(np.log2(float(step))), "a step is a power of two"
    assert n_nodes == sh.info["n_nodes"]
    assert parents[0] == 0 and (parents[1:] <= 1).all()
    used = parents > 0
    used[0] = True
    # every record item holds a record: its geometry word names an existing geometry
    rec = items[used & ~is_node]
    assert rec.shape[0] == n_records == sh.info["n_records"] > 0
    assert ((rec[:, 14] & 0x00FFFFFF) < sh.geometry_count()).all()
    # what is neither node nor record is padding, and stays zero
    assert not items[~used].any()


@pytest.mark.parametrize("n_tri,n_sph", [(1, 0), (0, 1), (2, 0), (1, 1), (3, 0), (2, 2), (5, 1), (9, 3)])
def test_tiny_scenes_on_host_match_brute_force(oracle, hostsim, n_tri, n_sph):
    """Trees whose root has one, two, three or four children (and the first two-level trees): hits equal brute force."""
    from tests.helpers import insert_tiny_scene, tiny_scene_rays
    so, sh = oracle.OracleScene(), hostsim.HostSimScene()
    for s in (so, sh):
        insert_tiny_scene(s, 0, n_tri, n_sph)
        s.commit()
    rays = tiny_scene_rays(oracle.RAY_DTYPE)
    want = so.intersect(rays, oracle.ISECT_BRUTE)
    assert (want["geom_id"] != 0xFFFFFFFF).mean() > 0.05
    assert sh.intersect(rays).tobytes() == want.tobytes()


def test_triangle_pairing_keeps_every_triangle_exact(oracle, hostsim):
    """Records hold two triangles that share an edge.  Whatever the pairing does with rotated, reversed, duplicate,
    non-manifold and degenerate triangles, every hit (distance, barycentrics, geometry and primitive id, normal) must
    equal the brute-force oracle's, bit for bit."""
    from tests.helpers import insert_pairing_torture, torture_rays
    so, sh = oracle.OracleScene(), hostsim.HostSimScene()
    for s in (so, sh):
        insert_pairing_torture(s)
        s.commit()
    assert sh.info["n_prims"] == so.prim_count()
    rays = torture_rays(oracle.RAY_DTYPE)
    got = sh.intersect(rays)
    want = so.intersect(rays, oracle.ISECT_BRUTE)
    assert (want["geom_id"] != 0xFFFFFFFF).mean() > 0.5
    assert set(np.unique(want["geom_id"]).tolist()) >= {0, 3, 4}
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("mesh,gain", [("mixed", None), ("mixed", "1.3"), ("needles", "1.3")])
def test_spatial_splits_keep_every_hit(oracle, hostsim, monkeypatch, mesh, gain):
    """The builder may cut a record at a plane and keep it once per part (spatial splits, what the reference asks of
    Embree with RTCBuildQuality::HIGH, crates/rt/src/aggregate/embree.rs:33-34).  A cut tree must give the brute-force
    hits bit for bit, like the uncut one; on the mixed-scale mesh the cuts pay by the builder's own cost estimate
    (default settings) and rays test a third fewer records; RTMI_SPATIAL_GAIN above 1 forces cuts that do not, so that
    heavily cut trees are covered too."""
    from tests.helpers import cube_rays, mixed_scale_mesh, needle_mesh
    v, t = mixed_scale_mesh() if mesh == "mixed" else needle_mesh()
    rays = cube_rays(oracle.RAY_DTYPE)
    so = oracle.OracleScene()
    so.insert_mesh(0, v, t)
    so.insert_sphere(0, (1.0, -2.0, 0.5), 1.5)
    so.commit()
    want = so.intersect(rays, oracle.ISECT_BRUTE)
    assert (want["geom_id"] != 0xFFFFFFFF).mean() > 0.05
    stats = {}
    for spatial in ("0", "1"):
        monkeypatch.setenv("RTMI_SPATIAL", spatial)
        if gain is not None:
            monkeypatch.setenv("RTMI_SPATIAL_GAIN", gain)
        sh = hostsim.HostSimScene()
        sh.insert_mesh(0, v, t)
        sh.insert_sphere(0, (1.0, -2.0, 0.5), 1.5)
        sh.commit()
        assert sh.intersect(rays).tobytes() == want.tobytes()
        assert sh.stack_high() <= sh.info["stack_need"]
        stats[spatial] = (sh.info, sh.visits(rays))
    plain, cut = stats["0"], stats["1"]
    assert plain[0]["spatial_splits"] == 0 and plain[0]["n_records"] == t.shape[0] + 1  # no two triangles share an edge
    assert cut[0]["spatial_splits"] > (10 if gain is None else 100)
    assert cut[0]["n_records"] > plain[0]["n_records"]
    if mesh == "mixed":
        assert cut[1][1] < 0.8 * plain[1][1], "fewer record tests per ray"
        assert cut[1][0] < plain[1][0], "and fewer node visits"


def test_exr_files_read_back(tmp_path):
    """crates/rt-cli/src/output/file_output.rs:28-37 saves every channel as <name>.exr, RGB in 32-bit floats, luma
    channels replicated into the three.  The CLI's writer against an independent reader of the format: every float
    comes back bit for bit, NaN, infinities and denormals included."""
    from raytracing_amd import cli
    from tests.helpers import read_exr
    rng = np.random.default_rng(4)
    img = rng.normal(size=(37, 53, 3)).astype(np.float32)
    img[0, 0] = (np.nan, np.inf, -np.inf)
    img[1, 1] = (1e-42, -0.0, 3.4e38)
    cli.write_exr(tmp_path / "Color.exr", img)
    got = read_exr(tmp_path / "Color.exr")
    assert sorted(got) == ["B", "G", "R"]
    for k, n in enumerate("RGB"):
        assert got[n].view(np.uint32).tobytes() == np.ascontiguousarray(img[:, :, k]).view(np.uint32).tobytes()
    z = rng.uniform(0, 5, (9, 16)).astype(np.float32)
    cli.write_exr(tmp_path / "Z.exr", z)
    got = read_exr(tmp_path / "Z.exr")
    assert all((got[n] == z).all() for n in "RGB")


def test_host_types():
    from raytracing_amd import types
    r = types.RenderRange.parse("1..7x4..45")
    assert (r.x, r.y) == (range(1, 7), range(4, 45))
    r = types.RenderRange.parse("1x4")
    assert (r.x, r.y) == (range(1, 2), range(4, 5))
    with pytest.raises(ValueError):
        types.RenderRange.parse("1x2x3")
    assert types.parse_spp("7..84") == range(7, 84) and types.parse_spp("32") == range(0, 32)
    assert types.parse_dimensions("800x600") == (800, 600)
    q = types.quat_from_axis_angle((0, 1, 0), np.pi)
    assert abs(q[1] - 1.0) < 1e-6 and abs(q[3]) < 1e-6
    t = types.Transform(translation=(1, 0, 0), scale=(2, 2, 2), rot=q)
    assert np.allclose(t.apply_points([[1, 1, 1]]), [[-1, 2, -2]], atol=1e-5)  # rot, then scale, then translation


# ------------------------------------------------------------------------------------------------ multi-rank

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, W, H, spp, out_path):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from raytracing_amd import api, scenes, types
    from raytracing_amd import distributed as D
    so = O.OracleScene()
    scenes.CornellBoxScene.insert_into(so)
    so.commit()
    cam = types.Camera.from_dimensions(W, H)
    tiler = api.Tiler(0, 0, W, H, 32, 32)
    xy = D.tile_pixels(tiler, D.partition_tiles(tiler.tile_count(), world, rank))
    acc, _ = so.render(cam, O.make_params(spp=spp, threads=2), xy)  # the oracle stands in for the GPU renderer
    frame = D.gather_frame(torch.from_numpy(acc), xy, W, H)
    # the reusable form, two frames in a row into one buffer: the second frame must replace the first
    g = D.FrameGatherer(xy, W, H, "cpu")
    buf = torch.zeros((22, W * H)) if rank == 0 else None
    g.gather(torch.from_numpy(acc) * 2.0, buf)
    again = g.gather(torch.from_numpy(acc), buf)
    if rank == 0:
        assert again is buf and torch.equal(again, frame)
        np.save(out_path, frame.numpy())
    else:
        assert frame is None and again is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_partition_and_gather(oracle, tmp_path):
    """world_size 2 over gloo: interleaved tiles, gather to rank 0; the assembled frame equals a one-process render bit
    for bit (ragged: 100x70 gives clipped edge tiles and unequal pixel counts per rank)."""
    import torch
    import torch.multiprocessing as mp
    from raytracing_amd import api, scenes, types
    from raytracing_amd import distributed as D
    W, H, spp = 100, 70, 3
    out = str(tmp_path / "frame.npy")
    mp.spawn(_rank_main, args=(2, _free_port(), W, H, spp, out), nprocs=2, join=True)
    frame = np.load(out)
    so = oracle.OracleScene()
    scenes.CornellBoxScene.insert_into(so)
    so.commit()
    cam = types.Camera.from_dimensions(W, H)
    want, _ = so.render(cam, oracle.make_params(spp=spp), full_frame_xy(W, H))
    assert frame.shape == (22, W * H) and same_bits(frame, want).all()
    # partition properties
    tiler = api.Tiler(0, 0, W, H, 32, 32)
    parts = [list(D.partition_tiles(tiler.tile_count(), 3, r)) for r in range(3)]
    assert sorted(sum(parts, [])) == list(range(tiler.tile_count()))
    assert max(map(len, parts)) - min(map(len, parts)) <= 1
    # block order of a tile's pixels: the same pixels, 64 consecutive entries forming an 8 x 8 block where the tile allows
    blocked = D.tile_pixels(tiler, range(tiler.tile_count()), (8, 8))
    plain = D.tile_pixels(tiler, range(tiler.tile_count()))
    assert sorted(map(tuple, blocked.tolist())) == sorted(map(tuple, plain.tolist()))
    first = blocked[:64]
    assert first[:, 0].max() - first[:, 0].min() == 7 and first[:, 1].max() - first[:, 1].min() == 7
    # single-rank path of gather_frame (no process group)
    xy = D.tile_pixels(tiler, range(tiler.tile_count()))
    acc, _ = so.render(cam, oracle.make_params(spp=spp), xy)
    one = D.gather_frame(torch.from_numpy(acc), xy, W, H).numpy()
    assert same_bits(one, want).all()


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` outside a torch.distributed launch starts two ranks itself; with --dry-cpu --backend
    gloo they run everything around the GPU work (rendezvous, tile partition, gather, timer reduction, one JSON line
    from rank 0) on this machine, and a failing rank makes the launcher fail."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-cpu",
           "--backend", "gloo", "--width", "150", "--height", "85"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_cpu"] and out["frame_ok"] and out["pixels"] == 150 * 85 and out["tiles"] == 15
    # without --dry-cpu the ranks need GPUs: here they must fail, and the launcher must say so with its exit code
    if not has_gpu():
        r = subprocess.run(cmd[:8], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode != 0


def build_c_host(tmp_path):
    """examples/abi_smoke.c, a host in plain C99 against include/rtmi.h and librtmi.so alone."""
    import subprocess
    exe = str(tmp_path / "abi_smoke")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "abi_smoke.c"), "-L" + os.path.join(ROOT, "raytracing_amd"),
                        "-lrtmi", "-lm", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "raytracing_amd") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    return exe, env


def test_header_is_plain_c_and_a_c_host_links(tmp_path):
    """The boundary is a C ABI: the header compiles as strict C99 and a C program links against the library with no
    C++ or Python in between.  Without a GPU the program must stop at its first call with the library's error."""
    import subprocess
    exe, env = build_c_host(tmp_path)
    if has_gpu():
        pytest.skip("a GPU is present: the run is covered by the GPU tests")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=60)
    assert r.returncode == 1 and "no HIP device" in r.stderr
