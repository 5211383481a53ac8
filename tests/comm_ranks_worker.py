"""Several ranks of rtmi_film_gather on ONE GPU: one thread per rank, each with its own rtmi_device, scene and
communicator, through the RCCL stand-in of tests/fake_rccl (RCCL itself refuses two ranks on one device).

Run as a program by tests/test_gpu_comm_ranks.py, in a process of its own, because the library opens its collective
library once per process (RTMI_RCCL_LIB must be set before the first communicator is made).  Prints one "ok <name>" line
per scenario and exits 0, or prints what went wrong and exits 1.  Reference semantics: tiles are disjoint and nothing is
reduced (crates/rt-cli/src/executor.rs:307-330).
"""
import os
import sys
import threading
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "_build", "libfake_rccl.so")
os.environ["RTMI_RCCL_LIB"] = FAKE
os.environ.setdefault("FAKE_RCCL_TIMEOUT_S", "20")

import numpy as np  # noqa: E402

from raytracing_amd import api, scenes, types  # noqa: E402
from raytracing_amd import distributed as D  # noqa: E402
from raytracing_amd._lib import RtmiError  # noqa: E402

SPP = 2
JOIN_S = 120


def same_bits(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


class Rank:
    """What one rank owns: a device handle, the scene on it, a communicator."""

    def __init__(self, rank, world, ident):
        self.rank, self.world = rank, world
        self.device = api.Device(0)
        sg = api.HipScene(self.device)
        scenes.CornellBoxScene.insert_into(sg)
        self.scene = sg
        self.world_scene = sg.commit().into_world()
        self.comm = api.Comm(self.device, ident, rank, world)

    def film_of_tiles(self, tiler, tiles, cam):
        film = api.Film(self.device, D.tile_pixels(tiler, tiles, (8, 8)))
        api.render(self.world_scene, film, cam, SPP)
        return film

    def close(self):
        self.comm.close()
        self.scene.close()
        self.device.close()


def run_ranks(world, body):
    """body(rank_index, ident) on `world` threads; returns their results, raises the first exception."""
    ident = api.Comm.unique_id()
    results, errors = [None] * world, [None] * world

    def main(r):
        try:
            results[r] = body(r, ident)
        except BaseException:  # noqa: BLE001 - reported by the caller
            errors[r] = traceback.format_exc()

    threads = [threading.Thread(target=main, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(JOIN_S)
    if any(t.is_alive() for t in threads):
        print("FAIL: a rank is still running after %d s (ranks out of step)" % JOIN_S, flush=True)
        os._exit(1)
    for r, e in enumerate(errors):
        if e:
            raise RuntimeError("rank %d:\n%s" % (r, e))
    return results


def reference_frame(W, H, cam):
    dev = api.Device(0)
    sg = api.HipScene(dev)
    scenes.CornellBoxScene.insert_into(sg)
    world_scene = sg.commit().into_world()
    ys, xs = np.mgrid[0:H, 0:W]
    film = api.Film(dev, np.stack([xs.ravel(), ys.ravel()], 1).astype(np.uint32))
    api.render(world_scene, film, cam, SPP)
    acc = film.download()
    film.close()
    sg.close()
    dev.close()
    return acc


def scenario_gather(world, root, W, H):
    """Every rank renders its interleaved tiles; the gathered frame equals the frame rendered at once."""
    cam = types.Camera.from_dimensions(W, H)
    tiler = api.Tiler(0, 0, W, H, 32, 32)
    want = reference_frame(W, H, cam)

    def body(r, ident):
        me = Rank(r, world, ident)
        film = me.film_of_tiles(tiler, api.tile_partition(tiler.tile_count(), world, r), cam)
        frame = api.Film.frame(me.device, W, H) if r == root else None
        out = []
        for _ in range(2):  # the second gather reuses the plan: no list exchange
            if frame is not None:
                frame.clear()
            film.gather(me.comm, frame, root)
            if frame is not None:
                out.append(frame.download())
        n = film.n_pixels
        film.close()
        if frame is not None:
            frame.close()
        me.close()
        return n, out

    res = run_ranks(world, body)
    assert sum(n for n, _ in res) == W * H
    for got in res[root][1]:
        assert same_bits(got, want).all(), "gathered frame differs from the frame rendered at once"
    return [n for n, _ in res]


def scenario_film_switch():
    """Rank 1 alternates between two films of equal size whose pixel lists differ while the root reuses one film: the
    lists must travel again each time, or the planes land on the wrong pixels."""
    W, H, world = 128, 32, 2
    cam = types.Camera.from_dimensions(W, H)
    tiler = api.Tiler(0, 0, W, H, 32, 32)
    want = reference_frame(W, H, cam)

    def body(r, ident):
        me = Rank(r, world, ident)
        out = []
        if r == 0:
            film = me.film_of_tiles(tiler, [0, 2], cam)
            frame = api.Film.frame(me.device, W, H)
            for _ in range(4):
                frame.clear()
                film.gather(me.comm, frame, 0)
                out.append(frame.download())
            frame.close()
            film.close()
        else:
            a = me.film_of_tiles(tiler, [1, 3], cam)
            b = me.film_of_tiles(tiler, [3, 1], cam)  # same pixels, other order
            assert a.n_pixels == b.n_pixels and not np.array_equal(a.xy, b.xy)
            for film in (a, b, a, a):
                film.gather(me.comm, None, 0)
            a.close()
            b.close()
        me.close()
        return out

    res = run_ranks(world, body)
    for k, got in enumerate(res[0]):
        assert same_bits(got, want).all(), "gather %d after a change of film put planes on the wrong pixels" % k


def scenario_errors():
    """A rank that cannot take part makes every rank return an error (nobody waits), and the next good gather works."""
    W, H, world = 128, 32, 2
    cam = types.Camera.from_dimensions(W, H)
    tiler = api.Tiler(0, 0, W, H, 32, 32)
    want = reference_frame(W, H, cam)

    def body(r, ident):
        me = Rank(r, world, ident)
        film = me.film_of_tiles(tiler, api.tile_partition(4, world, r), cam)
        frame = api.Film.frame(me.device, W, H) if r == 0 else None
        small = api.Film.frame(me.device, 64, 32) if r == 0 else None
        log = []

        def attempt(name, f, fr):
            try:
                f.gather(me.comm, fr, 0)
                log.append((name, None))
            except RtmiError as exc:
                log.append((name, str(exc)))

        # 1. rank 1 brings a film with a pixel outside the frame
        bad = api.Film(me.device, np.array([[W + 5, 3]], dtype=np.uint32)) if r == 1 else film
        attempt("pixel outside the frame", bad, frame)
        # 2. the root passes a frame that is too small for the films
        attempt("frame too small", film, small)
        # 3. the root passes no frame at all
        attempt("no frame", film, None)
        # 4. rank 1 gives up its turn
        if r == 1:
            me.comm.gather_abort()
            log.append(("abort", "gave up"))
        else:
            attempt("abort", film, frame)
        # 5. and now a good one
        if frame is not None:
            frame.clear()
        attempt("good", film, frame)
        got = frame.download() if frame is not None else None
        for f in (film, frame, small, bad if bad is not film else None):
            if f is not None:
                f.close()
        me.close()
        return log, got

    res = run_ranks(world, body)
    for r in range(world):
        log = dict(res[r][0])
        for name in ("pixel outside the frame", "frame too small", "no frame", "abort"):
            assert log[name] is not None, "rank %d: '%s' did not fail: %r" % (r, name, log)
        assert log["good"] is None, "rank %d: the gather after the failed ones failed too: %s" % (r, log["good"])
    assert same_bits(res[0][1], want).all()
    return {r: res[r][0] for r in range(world)}


def main():
    if not os.path.exists(FAKE):
        print("FAIL: %s is not built (make -C tests/fake_rccl)" % FAKE)
        return 1
    try:
        print("ok gather world=2: pixels per rank %s" % scenario_gather(2, 0, 96, 80), flush=True)
        print("ok gather world=3, root=1: pixels per rank %s" % scenario_gather(3, 1, 96, 80), flush=True)
        # two tiles over three ranks: rank 2 owns nothing, and it is the root
        print("ok gather with a root that owns no pixels: %s" % scenario_gather(3, 2, 64, 32), flush=True)
        scenario_film_switch()
        print("ok film switch", flush=True)
        logs = scenario_errors()
        print("ok errors reach every rank: %s" % {r: [(n, (m or "")[:60]) for n, m in logs[r]] for r in logs}, flush=True)
    except Exception:  # noqa: BLE001
        traceback.print_exc()
        print("FAIL", flush=True)
        return 1
    print("ALL OK", flush=True)
    return 0


if __name__ == "__main__":
    sys.stdout.flush()
    os._exit(main())
