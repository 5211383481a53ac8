#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (bounces included) and ms/frame for the dragon scene, 1920x1080, 100 spp.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one whole frame: every rank (one process per GPU) traces all samples of its interleaved share of the 32x32
tiles into accumulators resident in HBM, then the accumulator planes are gathered to rank 0 (RCCL).  The scene
(mesh, BVH) is built and uploaded before the timed region.  Rank 0 prints one JSON line.

The dragon mesh the reference uses is not distributed with it; unless assets/obj/dragon.obj exists the scene uses the
deterministic 5,087,232-triangle stand-in of raytracing_amd/standin.py (config.mesh says which).
"""
import argparse
import json
import os
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


def parse_args():
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
    return ap.parse_args()


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


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RTMI_BENCH_DEVICE" in os.environ:  # rehearsal of the multi-rank path on a box with fewer GPUs than ranks
        local_rank = int(os.environ["RTMI_BENCH_DEVICE"])
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the only path")
    torch.cuda.set_device(local_rank)
    dev_t = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev_t)

    from raytracing_amd import api, types
    from raytracing_amd import distributed as D

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

    # RTMI_BENCH_SHARE=N: render only the share of tiles one rank of an N-rank job would own (scaling rehearsal on 1 GPU)
    share = int(os.environ.get("RTMI_BENCH_SHARE", "0"))
    my_tiles = D.partition_tiles(tiler.tile_count(), share, 0) if share > 1 and world == 1 else \
        D.partition_tiles(tiler.tile_count(), world, rank)
    block = tuple(int(v) for v in os.environ.get("RTMI_BENCH_BLOCK", "8x8").split("x"))
    xy = D.tile_pixels(tiler, my_tiles, block if block[0] > 0 else None)
    accum = torch.zeros((api.ACCUM_PLANES, xy.shape[0]), dtype=torch.float32, device=dev_t)
    torch.cuda.synchronize()
    film = api.Film(device, xy, accum.data_ptr())

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    gatherer = D.FrameGatherer(xy, W, H, dev_t)
    frame_buf = torch.zeros((api.ACCUM_PLANES, W * H), dtype=torch.float32, device=dev_t) if rank == 0 else None

    def step():
        accum.zero_()
        torch.cuda.synchronize()
        st = api.render(world_scene, film, camera, args.spp, None, args.max_ray_depth, 0, args.seed,
                        args.samples_per_pass)
        frame = gatherer.gather(accum, frame_buf)
        return st, frame

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    rays = paths = 0
    gpu_ms = bounce_ms = extend_ms = 0.0
    passes = launches = 0
    for _ in range(args.steps):
        st, frame = step()
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
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_cnt, op=dist.ReduceOp.SUM)
    elapsed = float(t_el.item())
    total_rays, total_paths = int(t_cnt[0].item()), int(t_cnt[1].item())

    if rank == 0:
        steps = max(args.steps, 1)
        ms_per_step = elapsed / steps * 1e3
        value = total_rays / elapsed / 1e6
        # roofline of the dominant kernel (extend_kernel: BVH traversal + primitive tests of every ray of a bounce),
        # from this rank's own launches: HIP events around every launch give its time, a separate untimed render
        # with the visit counters on gives the node and record fetches
        roofline = None
        if not args.no_visits:
            device.count_visits(True)
            film_c = api.Film(device, xy)
            st_c = api.render(world_scene, film_c, camera, args.spp, None, args.max_ray_depth, 0, args.seed,
                              args.samples_per_pass)
            nodes, records = device.visits()
            device.count_visits(False)
            film_c.close()
            n_launch = max(launches // steps, 1)
            # the last ~1 % of the rays of a frame are traced by tail_kernel, which neither counts visits nor is timed here
            extend_rays = st_c["rays"] - st_c["tail_rays"]
            alg_bytes = nodes * NODE_BYTES + records * RECORD_BYTES + extend_rays * RAY_BYTES
            kernel_s = extend_ms / steps / 1e3
            achieved = alg_bytes / kernel_s / 1e9 if kernel_s > 0 else 0.0
            traffic = None
            tfile = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tfile):
                try:
                    tj = json.load(open(tfile))
                    key = "%s_%dx%d_%dspp" % (args.scene, W, H, args.spp)
                    if key in tj:
                        traffic = tj[key]["hbm_bytes_per_launch"]
                except Exception:
                    traffic = None
            scene_bytes = info["node_bytes"] + info["prim_bytes"]
            residency = "L2 (4 MB per XCD)" if scene_bytes < 4e6 else \
                "infinity cache (256 MB)" if scene_bytes < 128e6 else "hbm + infinity cache"
            roofline = {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": "extend_kernel",
                "launches_per_step": n_launch,
                "avg_launch_ms": extend_ms / max(launches, 1),
                "kernel_ms_per_step": extend_ms / steps,
                "other_kernels_ms_per_step": (gpu_ms - extend_ms) / steps,
                "algorithmic_bytes_per_step": alg_bytes,
                "algorithmic_bytes_per_launch": alg_bytes / n_launch,
                "rays_per_step": extend_rays,
                "tail_kernel_rays_per_step": st_c["tail_rays"],
                "nodes_per_ray": nodes / max(extend_rays, 1),
                "records_per_ray": records / max(extend_rays, 1),
                "bytes_per_ray": alg_bytes / max(extend_rays, 1),
                "scene_resident_in": residency,
                "note": "scene: %.1f MB nodes + %.1f MB primitive records (64-B node / 64-B record gathers), resident in %s%s"
                        % (info["node_bytes"] / 1e6, info["prim_bytes"] / 1e6, residency,
                           "" if residency == "hbm + infinity cache" else
                           ": the kernel is not HBM-bound on this scene and the fraction of HBM peak says little"),
            }
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
        print(json.dumps(out))
    film.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
