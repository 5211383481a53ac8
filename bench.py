#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (bounces included) and ms/frame for the dragon scene, 1920x1080, 100 spp.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Run plainly with --gpus N > 1 it starts the N ranks itself (the second command line, as a child process) and passes
their output through.  A step is one whole frame: every rank (one process per GPU) traces all samples of its
interleaved share of the 32x32 tiles into a film resident in HBM, then the per-pixel accumulators are gathered into
rank 0's frame (rtmi_film_gather: RCCL send/receive over xGMI; a local scatter for one rank).  The scene (mesh, BVH) is
built and uploaded before the timed region.  Rank 0 prints one JSON line.  torch.distributed is used for the launch
contract only: rendezvous, barriers, the max / sum of the timers, handing out the communicator id.

The dragon mesh the reference uses is not distributed with it; unless assets/obj/dragon.obj exists the scene uses the
deterministic 5,087,232-triangle stand-in of raytracing_amd/standin.py (config.mesh says which).

--dry-cpu --backend gloo rehearses everything around the GPU work on a machine without GPUs (launcher, rendezvous,
tile partition, gather of synthetic accumulators, timer reduction, the JSON line); it measures nothing.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# algorithmic bytes of the traversal kernel: one 64-B BVH4 node per node visit, one 64-B
# primitive record (up to two triangles) per leaf visit, and per ray the ray read (origin + direction, 24 B) and the
# hit record written (t, Ng, geometry word, 20 B)
NODE_BYTES = 64
RECORD_BYTES = 64
RAY_BYTES = 24 + 20
HBM_PEAK_GBS = 8000.0
LINE_BYTES = 128
# Scattered reads that miss L2 complete at one rate on this GPU whether 64 or 128 bytes of the line are used, from the
# Infinity Cache and from HBM alike: 55.9 G lines/s (profiles/experiments/fetch_calib.hip, gather<64> / gather<128> on a
# 512 MiB table; profiles/r03_pmc/fetch_calib.txt).  One fabric read request (TCC_EA0_RDREQ) is one such line.
LINE_RATE_CEILING = 55.9e9


def csrc_hash():
    """sha256 over the library's sources: counter measurements (profiles/traffic.json) are only paired with timings of
    the build they were taken on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raytracing_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".cpp", ".h")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="dragon", choices=["dragon", "dragon-diffuse", "bunny", "cornell-box"])
    ap.add_argument("--spp", type=int, default=100)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--max-ray-depth", type=int, default=64)
    ap.add_argument("--tile-size", type=int, default=32)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--levels", type=int, default=5, help="subdivision levels of the stand-in dragon mesh")
    ap.add_argument("--standin-extent", type=float, default=None,
                    help="height of the stand-in mesh in dragon.obj units (default: calibrated to the reference render)")
    ap.add_argument("--samples-per-pass", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-visits", action="store_true", help="skip the counting render that sizes the roofline")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (gloo: --dry-cpu)")
    ap.add_argument("--dry-cpu", action="store_true", help="rehearse launch, partition and gather without a GPU")
    ap.add_argument("--gather", default="abi", choices=["abi", "torch"],
                    help="frame gather: the C ABI's (rtmi_film_gather) or torch.distributed on a torch-owned buffer")
    return ap.parse_args(argv)


def launch_ranks(args):
    """--gpus N > 1 outside a torch.distributed launch: start the N ranks as a child (never re-exec a process, and
    nothing here has touched a GPU), pass their output through, leave with their exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def insert_scene(name, scene, levels, extent=None):
    from raytracing_amd import scenes
    if name == "dragon":
        scenes.DragonScene.insert_into(scene, levels=levels, extent=extent)
    elif name == "dragon-diffuse":
        scenes.DragonScene.insert_into(scene, diffuse=True, levels=levels, extent=extent)
    else:
        scenes.AVAILABLE_SCENES[name].insert_into(scene)


def cpu_baseline(args, camera, tiler):
    """The CPU oracle (a port of the reference's algorithm, not Embree) on a bounded sample of the same workload:
    every 4th tile of the frame, as many leading samples as fit the time target."""
    from oracle import oracle as O
    from raytracing_amd import distributed as D
    t0 = time.time()
    so = O.OracleScene()
    insert_scene(args.scene, so, args.levels, args.standin_extent)
    so.commit()
    build_s = time.time() - t0
    xy = D.tile_pixels(tiler, range(0, tiler.tile_count(), 4))
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    prm = O.make_params(spp=args.spp, sample_begin=0, sample_end=1, max_depth=args.max_ray_depth, seed=args.seed,
                        threads=threads)
    t0 = time.time()
    _, rays1 = so.render(camera, prm, xy)
    dt1 = max(time.time() - t0, 1e-6)
    n = int(max(1, min(args.spp - 1, (args.cpu_seconds - dt1) / dt1)))
    prm = O.make_params(spp=args.spp, sample_begin=1, sample_end=1 + n, max_depth=args.max_ray_depth, seed=args.seed,
                        threads=threads)
    t0 = time.time()
    _, rays = so.render(camera, prm, xy)
    dt = max(time.time() - t0, 1e-6)
    return {
        "value": rays / dt / 1e6,
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": "every 4th 32x32 tile (%d px) x samples 1..%d of %d, %.1f s; oracle BVH build %.1f s not included"
                  % (xy.shape[0], n, args.spp, dt, build_s),
        "rays": int(rays),
        "seconds": dt,
    }


def dry_cpu(args, rank, world):
    """No GPU: the ranks partition the tiles, fill accumulators with a function of the pixel and the plane, gather them
    over gloo and reduce their timers, exactly as the measured path does around its GPU work."""
    import torch
    import torch.distributed as dist
    from raytracing_amd import api
    from raytracing_amd import distributed as D
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend)
    W, H = args.width, args.height
    tiler = api.Tiler(0, 0, W, H, args.tile_size, args.tile_size)
    my_tiles = api.tile_partition(tiler.tile_count(), world, rank)
    xy = D.tile_pixels(tiler, my_tiles, (8, 8))
    lin = torch.from_numpy(xy[:, 1].astype(np.int64) * W + xy[:, 0].astype(np.int64))
    planes = torch.arange(api.ACCUM_PLANES, dtype=torch.float32)[:, None]
    accum = planes * 1e6 + lin[None, :].to(torch.float32)
    gatherer = D.FrameGatherer(xy, W, H, "cpu")
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(max(args.steps, 1)):
        frame = gatherer.gather(accum)
    if world > 1:
        dist.barrier()
    t_el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    t_cnt = torch.tensor([xy.shape[0], len(my_tiles)], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_cnt, op=dist.ReduceOp.SUM)
    if rank == 0:
        want = planes * 1e6 + torch.arange(W * H, dtype=torch.float32)[None, :]
        ok = bool(torch.equal(frame, want)) and int(t_cnt[0]) == W * H and int(t_cnt[1]) == tiler.tile_count()
        emit(json.dumps({"metric": "dry run (no GPU work)", "value": 0.0, "unit": "Mrays/s", "n_gpus": world,
                         "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(t_el.item()) * 1e3,
                         "dry_cpu": True, "frame_ok": ok, "pixels": int(t_cnt[0]), "tiles": int(t_cnt[1])}))
        if not ok:
            raise SystemExit("dry run: the gathered frame is wrong")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


_result_fd = None


def keep_stdout_for_the_result():
    """The contract is one JSON line on stdout.  Libraries loaded here write banners of their own to file descriptor 1
    (RCCL prints its version table when a communicator is created), so everything written to it from now on goes to
    stderr, and emit() writes the line to what stdout was."""
    global _result_fd
    sys.stdout.flush()
    _result_fd = os.dup(1)
    os.dup2(2, 1)


def emit(line):
    data = (line + "\n").encode()
    if _result_fd is None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    else:
        os.write(_result_fd, data)


def make_roofline(args, device, world_scene, film, camera, info, profile, steps, extend_ms, gpu_ms, launches, world, share):
    """Roofline of the dominant kernel: the closest-hit launch of every bounce (BVH traversal + primitive tests;
    extend_queue_kernel on the first three bounces, extend_kernel on the others: one piece of work scheduled two ways).

    Times are this run's: HIP events around every launch on the library's stream, per bounce.  Bytes come from two
    places.  (1) Counters: profiles/traffic.json holds, per launch of one frame of this workload, the fabric read
    requests (TCC_EA0_RDREQ, one per 128-byte line: profiles/r03_pmc/fetch_calib.txt) and WRITE_SIZE from separate
    rocprofv3 --pmc passes, stamped with the hash of the sources they were measured on; they are used only when that hash
    is this build's.  `achieved` = (lines x 128 + bytes written) / time: what the memory side moves, whole lines counted,
    and `frac` = that / the HBM peak.  (2) The algorithmic count of the same launches (64 bytes per node visit and
    record test from device counters of an untimed render + ray and hit record), kept beside it as
    algorithmic_rate_gbs: it prices bytes L2 serves against HBM and can exceed the peak, so it is not the fraction."""
    from raytracing_amd import api
    W, H = args.width, args.height
    device.count_visits(True)
    film_c = api.Film(device, film.xy)
    st_c = api.render(world_scene, film_c, camera, args.spp, None, args.max_ray_depth, 0, args.seed,
                      args.samples_per_pass)
    nodes, records = device.visits()
    device.count_visits(False)
    film_c.close()
    n_launch = max(launches // steps, 1)
    # the bounces after the switch to tail_kernel (an eighth of a frame's rays on the headline workload) are traced
    # by that kernel, which counts no visits; it is reported in its own block below
    extend_rays = st_c["rays"] - st_c["tail_rays"]
    alg_bytes = nodes * NODE_BYTES + records * RECORD_BYTES + extend_rays * RAY_BYTES
    kernel_s = extend_ms / steps / 1e3
    alg_rate = alg_bytes / kernel_s / 1e9 if kernel_s > 0 else 0.0

    # per bounce, from this run's events
    per_bounce, tail = [], None
    for (depth, kernel), acc in sorted(profile.items()):
        row = {"depth": depth, "kernel": kernel, "rays": acc["rays"] // steps, "ms": acc["extend_ms"] / steps,
               "shade_ms": acc["shade_ms"] / steps}
        if kernel == "tail_kernel":
            tail = row
        else:
            per_bounce.append(row)

    # counters of the same launches, if they were taken on this build
    tj, stale = None, None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    key = "%s_%dx%d_%dspp" % (args.scene, W, H, args.spp)
    if os.path.exists(tfile) and world == 1 and share <= 1 and args.standin_extent is None:
        try:
            entry = json.load(open(tfile)).get(key)
            if entry and entry.get("csrc_sha256") == csrc_hash():
                tj = entry
            elif entry:
                stale = "profiles/traffic.json was measured on other sources (%s, this build %s)" % (
                    entry.get("csrc_sha256"), csrc_hash())
        except Exception as exc:  # noqa: BLE001
            stale = "profiles/traffic.json unreadable: %s" % exc
    traffic = achieved = None
    line_rate = valu = None
    if tj:
        by_depth = {(r["depth"], r["kernel"]): r for r in tj["extend"]}
        tot_bytes = tot_ms = deep_lines = deep_ms = 0.0
        for row in per_bounce:
            c = by_depth.get((row["depth"], row["kernel"]))
            if not c or row["ms"] <= 0:
                continue
            row["lines"] = c["rdreq"]
            row["hbm_bytes"] = c["rdreq"] * LINE_BYTES + c["write_kib"] * 1024.0
            row["gbs"] = row["hbm_bytes"] / row["ms"] / 1e6
            row["glines_per_s"] = c["rdreq"] / row["ms"] / 1e6
            for k in ("l2_hit", "valu_issue", "lane_util", "ta_busy"):
                if k in c:
                    row[k] = c[k]
            tot_bytes += row["hbm_bytes"]
            tot_ms += row["ms"]
            if row["depth"] >= 2:
                deep_lines += c["rdreq"]
                deep_ms += row["ms"]
        n_counted = len([r for r in per_bounce if "hbm_bytes" in r])
        if tail and tj.get("tail") and tail["ms"] > 0:
            # tail_kernel is the same traversal for the bounces after the switch (with the shading step inside it)
            tot_bytes += tj["tail"]["rdreq"] * LINE_BYTES + tj["tail"]["write_kib"] * 1024.0
            tot_ms += tail["ms"]
            n_counted += 1
        if tot_ms > 0:
            achieved = tot_bytes / tot_ms / 1e6
            traffic = tot_bytes / max(n_counted, 1)
        if deep_ms > 0:
            line_rate = {"bounces": "2 and deeper (the memory-bound ones)", "glines_per_s": deep_lines / deep_ms / 1e6,
                         "ceiling_glines_per_s": LINE_RATE_CEILING / 1e9,
                         "frac": deep_lines / deep_ms / 1e6 / (LINE_RATE_CEILING / 1e9),
                         "ceiling_from": "profiles/r03_pmc/fetch_calib.txt (scattered 64-B and 128-B reads: 55.9 G lines/s)"}
        shallow = [r for r in per_bounce if r["depth"] < 2 and "valu_issue" in r]
        if shallow:
            valu = {"bounces": "0 and 1",
                    "valu_issue_frac_of": "the VALU wave instructions per SIMD and cycle of a pure v_fma_f32 stream "
                                          "(0.431, profiles/r03_pmc/pk_fma_probe.txt)",
                    "valu_issue_frac": sum(r["valu_issue"] * r["ms"] for r in shallow) / sum(r["ms"] for r in shallow),
                    "lane_util": sum(r["lane_util"] * r["ms"] for r in shallow) / sum(r["ms"] for r in shallow)}
        if tail and tj.get("tail") and tail["ms"] > 0:
            c = tj["tail"]
            tail["lines"] = c["rdreq"]
            tail["hbm_bytes"] = c["rdreq"] * LINE_BYTES + c["write_kib"] * 1024.0
            tail["gbs"] = tail["hbm_bytes"] / tail["ms"] / 1e6
            tail["frac"] = tail["gbs"] / HBM_PEAK_GBS
            tail["glines_per_s"] = c["rdreq"] / tail["ms"] / 1e6
            for k in ("l2_hit", "valu_issue", "lane_util", "ta_busy"):
                if k in c:
                    tail[k] = c[k]
    if tail:
        tail["mrays_per_s"] = tail["rays"] / tail["ms"] / 1e3 if tail["ms"] > 0 else 0.0
        tail["share_of_frame_ms"] = tail["ms"] / (gpu_ms / steps) if gpu_ms > 0 else None
    scene_bytes = info["node_bytes"] + info["prim_bytes"]
    residency = "L2 (4 MB per XCD)" if scene_bytes < 4e6 else \
        "infinity cache (256 MB)" if scene_bytes < 128e6 else "hbm + infinity cache"
    basis = "counters" if achieved is not None else "algorithmic"
    if achieved is None:
        achieved = alg_rate
    return {
        "bound": "hbm",
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "frac_basis": "memory-side bytes of these launches by the counters (fabric read requests x 128 B + bytes written) / "
                      "their time, measured live" if basis == "counters" else
                      "ALGORITHMIC bytes (no counter file for this build and workload%s): prices bytes L2 serves against "
                      "HBM and can exceed 1" % ("; " + stale if stale else ""),
        "kernel": "the traversal launches of a frame: extend_queue_kernel (bounces 0-2), extend_kernel (the bounces up to "
                  "the switch), tail_kernel (all later bounces in one launch)" if basis == "counters" else
                  "extend_kernel + extend_queue_kernel (the traversal launch of every wavefront bounce)",
        "limited_by": "bounces 0-1: instruction issue (two thirds of the VALU rate of a pure multiply-add stream, at 0.67 "
                      "lane utilisation); deeper bounces: that and the rate at which the memory system completes scattered "
                      "128-byte line reads (line_rate below), of which a 64-byte node or record uses half",
        "launches_per_step": n_launch + (1 if tail and basis == "counters" else 0),
        "avg_launch_ms": (extend_ms / steps + (tail["ms"] if tail and basis == "counters" else 0.0)) /
                         (n_launch + (1 if tail and basis == "counters" else 0)),
        "kernel_ms_per_step": extend_ms / steps + (tail["ms"] if tail and basis == "counters" else 0.0),
        "other_kernels_ms_per_step": (gpu_ms - extend_ms) / steps - (tail["ms"] if tail and basis == "counters" else 0.0),
        "wavefront_launches_per_step": n_launch,
        "wavefront_ms_per_step": extend_ms / steps,
        # the algorithmic figures cover the wavefront launches only (tail_kernel counts no visits)
        "algorithmic_rate_gbs": alg_rate,
        "algorithmic_bytes_per_step": alg_bytes,
        "algorithmic_bytes_per_launch": alg_bytes / n_launch,
        "rays_per_step": extend_rays,
        "nodes_per_ray": nodes / max(extend_rays, 1),
        "records_per_ray": records / max(extend_rays, 1),
        "bytes_per_ray": alg_bytes / max(extend_rays, 1),
        "line_rate": line_rate,
        "valu": valu,
        "per_bounce": per_bounce,
        "tail": tail,
        "scene_resident_in": residency,
        "note": "scene: %.1f MB nodes + %.1f MB primitive records (64-B node / 64-B record gathers), resident in %s%s"
                % (info["node_bytes"] / 1e6, info["prim_bytes"] / 1e6, residency,
                   "" if residency == "hbm + infinity cache" else
                   ": the kernel is not HBM-bound on this scene and the fraction of HBM peak says little"),
    }


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    keep_stdout_for_the_result()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.dry_cpu:
        return dry_cpu(args, rank, world)
    if "RTMI_BENCH_DEVICE" in os.environ:  # rehearsal of the multi-rank path on a box with fewer GPUs than ranks
        local_rank = int(os.environ["RTMI_BENCH_DEVICE"])
    # torch first: its wheel carries its own HIP runtime and RCCL, and the library then shares them
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the only path")
    torch.cuda.set_device(local_rank)
    dev_t = torch.device("cuda", local_rank)
    # RTMI_BENCH_COMM=1: go through the process group and the RCCL communicator even with one rank (a rehearsal of the
    # multi-rank code on a one-GPU box: the root then sends its planes to itself through RCCL)
    use_dist = world > 1 or os.environ.get("RTMI_BENCH_COMM", "0") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        dist.init_process_group(args.backend, device_id=dev_t, rank=rank, world_size=world)

    from raytracing_amd import api, types

    W, H = args.width, args.height
    camera = types.Camera.from_dimensions(W, H)
    tiler = api.Tiler(0, 0, W, H, args.tile_size, args.tile_size)
    device = api.Device(local_rank)
    t0 = time.time()
    sg = api.HipScene(device)
    insert_scene(args.scene, sg, args.levels, args.standin_extent)
    t_insert = time.time() - t0
    world_scene = sg.commit().into_world()
    info = world_scene.info()
    mesh = getattr(sg, "dragon_mesh", args.scene)

    # RTMI_BENCH_SHARE=N [RTMI_BENCH_SHARE_RANK=r]: render only the share of tiles rank r (default 0) of an N-rank job
    # would own (scaling rehearsal on 1 GPU)
    share = int(os.environ.get("RTMI_BENCH_SHARE", "0"))
    share_rank = int(os.environ.get("RTMI_BENCH_SHARE_RANK", "0"))
    my_tiles = api.tile_partition(tiler.tile_count(), share, share_rank) if share > 1 and world == 1 else \
        api.tile_partition(tiler.tile_count(), world, rank)
    block = tuple(int(v) for v in os.environ.get("RTMI_BENCH_BLOCK", "8x8").split("x"))
    film = api.Film.for_tiles(device, tiler, my_tiles, block if block[0] > 0 else None)
    frame = api.Film.frame(device, W, H) if rank == 0 else None

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # the frame gather: the library's own (RCCL through the C ABI), or torch.distributed on a copy of the accumulators
    comm = None
    gather_kind = "rtmi_film_gather (RCCL send/recv to rank 0)" if use_dist else "rtmi_film_gather (local scatter)"
    gatherer = frame_t = None
    gather_fallback = None  # set when the library's gather could not be used: the JSON line is then marked invalid

    def use_torch_gather():
        nonlocal gatherer, frame_t
        import ctypes
        from raytracing_amd import distributed as D
        accum_t = torch.zeros((api.ACCUM_PLANES, film.n_pixels), dtype=torch.float32, device=dev_t)
        frame_t = torch.zeros((api.ACCUM_PLANES, W * H), dtype=torch.float32, device=dev_t) if rank == 0 else None
        gatherer = D.FrameGatherer(film.xy, W, H, dev_t)
        hip = ctypes.CDLL("libamdhip64.so.7")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]

        def gather():
            torch.cuda.synchronize()
            if film.n_pixels:
                rc = hip.hipMemcpy(accum_t.data_ptr(), film.accum_device(), accum_t.numel() * 4, 3)  # device to device
                if rc != 0:
                    raise RuntimeError("hipMemcpy failed: %d" % rc)
            gatherer.gather(accum_t, frame_t)
            torch.cuda.synchronize()
        return gather

    if args.gather == "torch":
        gather_kind = "torch.distributed gather"
        gather = use_torch_gather()
    else:
        failure = ""
        try:
            if use_dist:
                ident = torch.zeros(api.COMM_ID_BYTES, dtype=torch.uint8, device=dev_t)
                if rank == 0:
                    ident.copy_(torch.frombuffer(bytearray(api.Comm.unique_id()), dtype=torch.uint8))
                dist.broadcast(ident, src=0)
                comm = api.Comm(device, ident.cpu().numpy().tobytes(), rank, world)
            film.gather(comm, frame)  # builds the exchange plan (pixel lists to rank 0) outside the timed region
        except Exception as exc:  # noqa: BLE001 - reported below, on every rank
            failure = "%s: %s" % (type(exc).__name__, exc)
        failed = torch.tensor([1 if failure else 0], dtype=torch.int32, device=dev_t)
        if use_dist:
            dist.all_reduce(failed, op=dist.ReduceOp.MAX)
        if int(failed.item()):
            # said loudly, and in the JSON line: the numbers below then include torch's gather, not the library's
            sys.stderr.write("[bench rank %d] rtmi_film_gather unusable (%s); using torch.distributed for the gather\n"
                             % (rank, failure or "another rank failed"))
            gather_kind = "torch.distributed gather (rtmi_film_gather failed: %s)" % (failure or "on another rank")
            gather_fallback = gather_kind
            gather = use_torch_gather()
        else:
            def gather():
                film.gather(comm, frame)

    def step():
        film.clear()
        st = api.render(world_scene, film, camera, args.spp, None, args.max_ray_depth, 0, args.seed,
                        args.samples_per_pass)
        gather()
        return st

    n_warm = max(args.warmup, 1) if world > 1 else args.warmup
    for _ in range(n_warm):
        step()
    if rank == 0 and n_warm > 0:
        # untimed check of the exchange on the warm-up frame: every pixel of the frame arrived with all its samples
        acc = frame.download() if gatherer is None else frame_t.cpu().numpy()
        complete = int((acc[21] == float(args.spp)).sum())
        expect = W * H if share <= 1 or world > 1 else film.n_pixels
        if complete != expect:
            raise SystemExit("bench: the gathered frame is incomplete (%d of %d pixels carry %d samples)" %
                             (complete, expect, args.spp))
        del acc
    barrier()
    t0 = time.perf_counter()
    rays = paths = 0
    gpu_ms = bounce_ms = extend_ms = 0.0
    passes = launches = 0
    profile = {}
    for _ in range(args.steps):
        st = step()
        for b in device.last_render_profile():
            acc = profile.setdefault((b["depth"], b["kernel"]), {"rays": 0, "extend_ms": 0.0, "shade_ms": 0.0, "launches": 0})
            for k in acc:
                acc[k] += b[k]
        rays += st["rays"]
        paths += st["paths"]
        gpu_ms += st["gpu_ms"]
        bounce_ms += st["bounce_ms"]
        extend_ms += st["extend_ms"]
        passes += st["passes"]
        launches += st["bounce_launches"]
    barrier()
    elapsed = time.perf_counter() - t0
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=dev_t)
    t_cnt = torch.tensor([rays, paths], dtype=torch.int64, device=dev_t)
    if use_dist:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_cnt, op=dist.ReduceOp.SUM)
    elapsed = float(t_el.item())
    total_rays, total_paths = int(t_cnt[0].item()), int(t_cnt[1].item())

    if rank == 0:
        steps = max(args.steps, 1)
        ms_per_step = elapsed / steps * 1e3
        value = total_rays / elapsed / 1e6
        # roofline of the dominant kernel (the traversal launches, one per bounce: BVH traversal + primitive tests of
        # every ray of the bounce; extend_queue_kernel on the first three bounces, extend_kernel on the rest, the same
        # work scheduled in two ways), from this rank's own launches: HIP events around every launch give its time, a separate untimed render
        # with the visit counters on gives the node and record fetches
        roofline = None
        if not args.no_visits:
            roofline = make_roofline(args, device, world_scene, film, camera, info, profile, steps, extend_ms, gpu_ms,
                                     launches, world, share)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, camera, tiler)
        out = {
            "metric": "Mrays/s (incl. bounces) + ms/frame, dragon 1080p 100spp at 1/2/4/8 MI355X",
            "value": value,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s %dx%d %d spp, path tracer, max depth %d, tile %d, seed %d" %
                            (args.scene, W, H, args.spp, args.max_ray_depth, args.tile_size, args.seed),
                "mesh": mesh,
                "triangles": int(info["n_prims"]),
                "bvh_nodes": int(info["n_nodes"]),
                "partition": "tiles interleaved over %d rank(s), full scene replica per GPU, gather to rank 0" % world,
                "gather": gather_kind,
            },
            "ms_per_frame": ms_per_step,
            "mpaths_per_s": total_paths / elapsed / 1e6,
            "rays_per_path": total_rays / max(total_paths, 1),
            "gpu_ms_per_step_rank0": gpu_ms / steps,
            "scene_build_ms": info["build_ms"],
            "scene_insert_ms": t_insert * 1e3,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if gather_fallback:
            # the exchange step measured is not the library's: not a number for this build's multi-GPU path
            out["valid"] = False
            out["invalid_reason"] = gather_fallback
        emit(json.dumps(out))
    film.close()
    if frame is not None:
        frame.close()
    if comm is not None:
        comm.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
