"""The closest-hit entry point that allocates nothing (rtmi_intersect_device_static) against the one it is a variant
of (rtmi_intersect_device), through the C ABI: same rays, same bytes out, on two calls in a row of which the second
is smaller, on a tree deep enough to use the statically allocated stack-overflow buffer and on a shallow one."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _soup(n, seed):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-1, 1, (n, 3))
    v = (c[:, None, :] + rng.normal(scale=1.5, size=(n, 3, 3))).astype(np.float32)
    return v.reshape(-1, 3), np.arange(3 * n, dtype=np.uint32).reshape(n, 3)


def _rays(dtype, m, seed):
    rng = np.random.default_rng(seed)
    rays = np.zeros(m, dtype=dtype)
    d = rng.normal(size=(m, 3))
    rays["dir"] = d / np.linalg.norm(d, axis=1)[:, None]
    rays["org"] = rng.uniform(-1, 1, (m, 3)) - 3 * rays["dir"]
    rays["tnear"] = 1e-5
    rays["tfar"] = np.inf
    rays["tfar"][::5] = 2.5
    return rays


@pytest.mark.parametrize("deep", [True, False])
def test_static_entry_point_matches_the_allocating_one(device, scene_classes, deep):
    from raytracing_amd import _lib, api
    from tests import hip_rt
    sg = api.HipScene(device)
    if deep:
        v, idx = _soup(60000, 11)
        sg.insert_mesh(0, v, idx)
    else:
        scene_classes["spheres"].insert_into(sg)
    world = sg.commit().into_world()
    assert (world.info()["stack_need"] > 24) == deep
    hit_bytes = _lib.HIT_DTYPE.itemsize
    sizes = (200000, 7001)  # the second call is smaller than the first
    batches = [_rays(_lib.RAY_DTYPE, m, 40 + k) for k, m in enumerate(sizes)]
    if not deep:
        for b in batches:
            b["org"] = b["org"] * 0.2
    d_rays = [hip_rt.DeviceBuffer.from_array(b) for b in batches]
    d_want = [hip_rt.DeviceBuffer(m * hit_bytes) for m in sizes]
    d_got = [hip_rt.DeviceBuffer(sizes[0] * hit_bytes) for _ in sizes]  # both as large as the first call's output
    stream = hip_rt.Stream()
    sp = C.c_void_p(stream.handle)
    for k, m in enumerate(sizes):
        world.intersect_device(d_rays[k].ptr, d_want[k].ptr, m, sp)
    stream.synchronize()
    want = [d_want[k].download(_lib.HIT_DTYPE) for k in range(2)]
    assert any((w["geom_id"] != 0xFFFFFFFF).any() for w in want)
    for b in d_got:
        b.fill(0xEE)
    free_before = hip_rt.free_bytes()
    for k, m in enumerate(sizes):
        world.intersect_device_static(d_rays[k].ptr, d_got[k].ptr, m, sp)
    stream.synchronize()
    assert hip_rt.free_bytes() == free_before, "the static entry point changed the amount of free device memory"
    for k, m in enumerate(sizes):
        got = d_got[k].download(_lib.HIT_DTYPE)
        assert got[:m].tobytes() == want[k].tobytes(), "call %d differs from rtmi_intersect_device" % k
        # nothing was written past the m hits the call was asked for
        assert (got[m:].view(np.uint8) == 0xEE).all()
    # and once more the other way round (large after small), on the default stream
    world.intersect_device_static(d_rays[0].ptr, d_got[1].ptr, sizes[0], None)
    hip_rt.ok(hip_rt.hip().hipDeviceSynchronize())
    assert d_got[1].download(_lib.HIT_DTYPE).tobytes() == want[0].tobytes()
    for b in d_rays + d_want + d_got:
        b.free()
    stream.destroy()


def test_static_entry_point_under_graph_capture(device):
    """What the entry point is for: captured into a HIP graph (global capture mode, under which a device allocation
    from any thread is an error), instantiated, and launched twice; every launch of the graph gives the hits of a
    direct call, on a tree deep enough to use the static stack-overflow buffer."""
    from raytracing_amd import _lib, api
    from tests import hip_rt
    hip = hip_rt.hip()
    sg = api.HipScene(device)
    v, idx = _soup(40000, 21)
    sg.insert_mesh(0, v, idx)
    world = sg.commit().into_world()
    assert world.info()["stack_need"] > 24
    m = 50000
    rays = _rays(_lib.RAY_DTYPE, m, 77)
    want = world.intersect(rays)
    d_rays = hip_rt.DeviceBuffer.from_array(rays)
    d_hits = hip_rt.DeviceBuffer(m * _lib.HIT_DTYPE.itemsize)
    stream = hip_rt.Stream()
    sp = C.c_void_p(stream.handle)
    hip_rt.ok(hip.hipStreamBeginCapture(sp, 0))  # hipStreamCaptureModeGlobal
    try:
        world.intersect_device_static(d_rays.ptr, d_hits.ptr, m, sp)
    finally:
        graph = C.c_void_p()
        rc = hip.hipStreamEndCapture(sp, C.byref(graph))
    hip_rt.ok(rc)
    assert graph.value
    gexec = C.c_void_p()
    hip_rt.ok(hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0))
    for _ in range(2):
        d_hits.fill(0xEE)  # nothing ran at capture time, and every launch writes every hit
        hip_rt.ok(hip.hipGraphLaunch(gexec, sp))
        stream.synchronize()
        assert d_hits.download(_lib.HIT_DTYPE).tobytes() == want.tobytes()
    hip_rt.ok(hip.hipGraphExecDestroy(gexec))
    hip_rt.ok(hip.hipGraphDestroy(graph))
    d_rays.free()
    d_hits.free()
    stream.destroy()


def test_static_entry_point_argument_errors(device, scene_classes):
    from raytracing_amd import api
    sg = api.HipScene(device)
    scene_classes["debug"].insert_into(sg)
    with pytest.raises(api.RtmiError) as e:
        api.CommittedHipScene(sg).intersect_device_static(1, 1, 4)  # not committed
    assert e.value.code == -3
    world = sg.commit().into_world()
    with pytest.raises(api.RtmiError) as e:
        world.intersect_device_static(0, 0, 4)
    assert e.value.code == -1
    world.intersect_device_static(0, 0, 0)  # nothing to do
