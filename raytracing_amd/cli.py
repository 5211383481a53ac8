"""Command line mirror of the reference's `rt-cli` (crates/rt-cli/src/main.rs:21-119, renderer.rs:14-98).

Same flags, same defaults:  python -m raytracing_amd.cli --scene cornell-box --spp 32 -d 800x600 -o file

  OutputBuffers.convert   output/mod.rs:23-60   RGB channels (Color, Normal, Position, Albedo) sRGB-encoded, Luma raw
  FileOutput              output/file_output.rs:21-59   output/ldr/<Channel>.jpeg and output/hdr/<Channel>.*
The HDR files are written as PFM and NPY (no EXR encoder is available here); `-o tev` is accepted and ignored (the
viewer streaming of output/tev_streaming.rs is not part of this build).
"""
from __future__ import annotations

import argparse
import os
import struct
import sys
import time

import numpy as np

from . import api, scenes, types

RGB_CHANNELS = ("Normal", "Position", "Albedo", "Color")
LUMA_CHANNELS = ("Variance", "Z", "RayDepth")


def build_parser():
    ap = argparse.ArgumentParser(prog="rt-cli (rtmi)", description=__doc__.split("\n\n")[0])
    ap.add_argument("tev_path", nargs="?", default=None)
    ap.add_argument("--spp", type=int, default=32)
    ap.add_argument("--sample-range", type=types.parse_spp, default=None)
    ap.add_argument("--scene", choices=list(scenes.AVAILABLE_SCENES), default="cornell-box")
    ap.add_argument("-d", "--dimensions", type=types.parse_dimensions, default=(800, 600))
    ap.add_argument("-o", "--output", action="append", choices=["tev", "file"], default=[])
    ap.add_argument("-i", "--integrator", choices=["basic", "path-tracer"], default="path-tracer")
    ap.add_argument("--tev-hostname", default=None)
    ap.add_argument("--allowed-error", type=float, default=None)
    ap.add_argument("--tile-size", type=int, default=32)
    ap.add_argument("-e", "--execution-mode", choices=["multithreaded", "monothreaded"], default="multithreaded")
    ap.add_argument("-r", "--range", type=types.RenderRange.parse, default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-ray-depth", type=int, default=None)
    ap.add_argument("--device", type=int, default=0, help="HIP device ordinal (not a reference flag)")
    ap.add_argument("--outdir", default="output", help="where FileOutput writes (the reference uses ./output)")
    return ap


class OutputBuffers:
    """GenericRenderResult<Rgb32FImage, Luma32FImage> (output/mod.rs:17-18): one image per channel."""

    def __init__(self, width, height):
        self.width, self.height = width, height
        self.channels = {name: np.zeros((height, width, 3), np.float32) for name in RGB_CHANNELS}
        self.channels.update({name: np.zeros((height, width), np.float32) for name in LUMA_CHANNELS})

    def convert_tile(self, msg: api.TileMsg):
        """convert() for every pixel of a TileMsg: RGB channels arrive sRGB-encoded (colorspace.rs:71-80 is applied on
        the GPU by the resolve kernel when the executor is asked for it), Luma channels raw."""
        t = msg.tile
        h, w = t.height(), t.width()
        for name, planes in api.CHANNELS.items():
            block = msg.data[:, planes].reshape(h, w, -1)
            if name in RGB_CHANNELS:
                self.channels[name][t.y_start:t.y_end, t.x_start:t.x_end] = block
            else:
                self.channels[name][t.y_start:t.y_end, t.x_start:t.x_end] = block[:, :, 0]


def _exr_attribute(name, kind, value):
    return name.encode() + b"\0" + kind.encode() + b"\0" + struct.pack("<i", len(value)) + value


def write_exr(path, img):
    """An OpenEXR file as the reference's `ImageBuffer<Rgb<f32>>::save("*.exr")` produces it for its readers
    (crates/rt-cli/src/output/file_output.rs:28-37): one part, scan lines, the three channels B, G, R as 32-bit floats;
    a one-channel image is replicated into the three (file_output.rs:24,35).  Written without compression, which every
    reader accepts (the image crate picks a lossless compression; the pixels are the same)."""
    img = np.asarray(img, dtype=np.float32)
    if img.ndim == 2:
        img = np.repeat(img[:, :, None], 3, 2)
    height, width = img.shape[:2]
    channels = b"".join(c + b"\0" + struct.pack("<iB3xii", 2, 0, 1, 1) for c in (b"B", b"G", b"R")) + b"\0"
    window = struct.pack("<4i", 0, 0, width - 1, height - 1)
    header = struct.pack("<ii", 20000630, 2)
    header += _exr_attribute("channels", "chlist", channels)
    header += _exr_attribute("compression", "compression", b"\0")
    header += _exr_attribute("dataWindow", "box2i", window)
    header += _exr_attribute("displayWindow", "box2i", window)
    header += _exr_attribute("lineOrder", "lineOrder", b"\0")
    header += _exr_attribute("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    header += _exr_attribute("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0))
    header += _exr_attribute("screenWindowWidth", "float", struct.pack("<f", 1.0))
    header += b"\0"
    row_bytes = 12 * width
    chunk = 8 + row_bytes
    first = len(header) + 8 * height
    offsets = (first + chunk * np.arange(height, dtype=np.uint64)).astype("<u8")
    body = np.zeros((height, chunk), dtype=np.uint8)
    body[:, 0:4] = np.arange(height, dtype="<i4").view(np.uint8).reshape(height, 4)
    body[:, 4:8] = np.full(height, row_bytes, dtype="<i4").view(np.uint8).reshape(height, 4)
    planes = np.ascontiguousarray(img[:, :, ::-1].transpose(0, 2, 1)).astype("<f4")  # per row: all B, all G, all R
    body[:, 8:] = planes.view(np.uint8).reshape(height, row_bytes)
    with open(path, "wb") as fh:
        fh.write(header)
        fh.write(offsets.tobytes())
        fh.write(body.tobytes())


class FileOutput:
    def __init__(self, outdir="output"):
        self.hdr_outdir = os.path.join(outdir, "hdr")
        self.ldr_outdir = os.path.join(outdir, "ldr")

    def commit(self, buffers: OutputBuffers):
        from PIL import Image
        os.makedirs(self.hdr_outdir, exist_ok=True)
        os.makedirs(self.ldr_outdir, exist_ok=True)
        for name, img in buffers.channels.items():
            write_exr(os.path.join(self.hdr_outdir, name + ".exr"), img)
            np.save(os.path.join(self.hdr_outdir, name + ".npy"), img)
            ldr = np.nan_to_num(img, nan=0.0, posinf=1.0, neginf=0.0)
            ldr = np.round(np.clip(ldr, 0.0, 1.0) * 255.0).astype(np.uint8)
            if ldr.ndim == 2:
                ldr = np.repeat(ldr[:, :, None], 3, 2)
            Image.fromarray(ldr, "RGB").save(os.path.join(self.ldr_outdir, name + ".jpeg"), quality=90)


class Renderer:
    """crates/rt-cli/src/renderer.rs:14-98."""

    def __init__(self, args):
        W, H = args.dimensions
        self.final_outputs = [FileOutput(args.outdir)] if "file" in args.output else []
        if "tev" in args.output:
            print("note: tev streaming is not part of this build; ignoring -o tev", file=sys.stderr)
        self.executor = api.Executor((W, H), tile_size=args.tile_size, allowed_error=args.allowed_error,
                                     integrator=args.integrator, spp=args.spp, seed=args.seed,
                                     max_depth=64 if args.max_ray_depth is None else args.max_ray_depth)
        self.execution_mode = args.execution_mode
        self.pixel_range = args.range or types.RenderRange(range(0, W), range(0, H))
        self.sample_range = args.sample_range if args.sample_range is not None else range(0, args.spp)

    def run(self, world):
        W, H = self.executor.dimension
        buffers = OutputBuffers(W, H)
        t0 = time.time()
        run = (self.executor.run_multithreaded if self.execution_mode == "multithreaded"
               else self.executor.run_monothreaded)
        stats = run(world, buffers.convert_tile, self.pixel_range, self.sample_range, srgb=True)
        dt = time.time() - t0
        print("run tile renderer: %.3f s (%.1f Mrays/s, %d rays, %d paths)" %
              (dt, stats["rays"] / max(dt, 1e-9) / 1e6, stats["rays"], stats["paths"]))
        for out in self.final_outputs:
            out.commit(buffers)
        return buffers, stats


def main(argv=None):
    args = build_parser().parse_args(argv)
    device = api.Device(args.device)
    print("loading scene")
    scene = api.HipScene(device)
    scenes.AVAILABLE_SCENES[args.scene].insert_into(scene)
    print("building scene")
    committed = scene.commit_with_progress(lambda amount: True)
    world = committed.into_world()
    info = world.info()
    print("scene: %d primitives, %d BVH nodes, built in %.0f ms" % (info["n_prims"], info["n_nodes"], info["build_ms"]))
    Renderer(args).run(world)
    return 0


if __name__ == "__main__":
    sys.exit(main())
