"""Host-side descriptors mirroring the reference's scene/camera/material API.

Same names and argument meaning as the reference types they stand for:
  DiffuseBxDF / DielectricBxDF / ThinDielectricBxDF / EmitBxDF   crates/rt/src/material/mod.rs:98-101,133-137,315-318,369-372
  MaterialDescriptor / LightDescriptor / MaterialId               crates/rt/src/material/mod.rs:401-430
  Transform                                                       crates/rt/src/math/transform.rs:6-10
  LookAt -> Quat                                                  crates/rt/src/math/quaternion.rs:7-27
  Camera                                                          crates/rt/src/camera.rs:9-57
  Dimensions / RenderRange / Spp                                  crates/rt-cli/src/utils/mod.rs:47-109,157-176
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

f32 = np.float32


@dataclass
class DiffuseBxDF:
    albedo: Sequence[float] = (0.0, 0.0, 0.0)


@dataclass
class DielectricBxDF:
    ior: float = 0.0
    roughness: float = 0.0


@dataclass
class ThinDielectricBxDF:
    ior: float = 0.0


@dataclass
class EmitBxDF:
    le: Sequence[float] = (0.0, 0.0, 0.0)


@dataclass
class MaterialDescriptor:
    label: Optional[str]
    material: object


@dataclass
class LightDescriptor:
    label: Optional[str]
    light_pos: Sequence[float]


def quat_from_axis_angle(axis, angle):
    """glam Quat::from_axis_angle: (axis * sin(angle/2), cos(angle/2)), evaluated in f32."""
    half = f32(angle) * f32(0.5)
    s, c = f32(np.sin(half)), f32(np.cos(half))
    return (float(f32(axis[0]) * s), float(f32(axis[1]) * s), float(f32(axis[2]) * s), float(c))


QUAT_IDENTITY = (0.0, 0.0, 0.0, 1.0)


@dataclass
class Transform:
    """translation + scale + rot (x, y, z, w); points get rot, then scale, then translation."""
    translation: Sequence[float] = (0.0, 0.0, 0.0)
    scale: Sequence[float] = (1.0, 1.0, 1.0)
    rot: Sequence[float] = QUAT_IDENTITY

    def apply_points(self, positions):
        """Transformer<Point>::apply (crates/rt/src/math/transform.rs:45-52) over an [n,3] array, in f32 with glam's
        operation order: q*v = v*(w*w - b.b) + b*(2 v.b) + (b x v)*(2w), then scale, then translation."""
        p = np.asarray(positions, dtype=np.float32)
        bx, by, bz, w = (f32(self.rot[0]), f32(self.rot[1]), f32(self.rot[2]), f32(self.rot[3]))
        b2 = f32(f32(f32(bx * bx) + f32(by * by)) + f32(bz * bz))
        k = f32(f32(w * w) - b2)
        w2 = f32(w * f32(2.0))
        x, y, z = p[:, 0], p[:, 1], p[:, 2]
        d2 = ((x * bx + y * by) + z * bz) * f32(2.0)
        cross = (by * z - y * bz, bz * x - z * bx, bx * y - x * by)
        out = np.empty_like(p)
        for i, (v, b) in enumerate(((x, bx), (y, by), (z, bz))):
            out[:, i] = f32(self.scale[i]) * ((v * k + b * d2) + cross[i] * w2) + f32(self.translation[i])
        return out


def look_at_quat(direction, forward):
    """crates/rt/src/math/quaternion.rs:12-27."""
    d = np.asarray(direction, dtype=np.float32)
    fw = np.asarray(forward, dtype=np.float32)
    d = d / f32(np.sqrt(np.dot(d, d)))
    fw = fw / f32(np.sqrt(np.dot(fw, fw)))
    cos = f32(np.dot(fw, d))
    angle = f32(np.arccos(np.clip(cos, -1.0, 1.0)))
    if abs(abs(float(cos)) - 1.0) > 0.01:
        axe = np.cross(fw, d).astype(np.float32)
        axe = axe / f32(np.sqrt(np.dot(axe, axe)))
        return quat_from_axis_angle(axe, angle)
    return quat_from_axis_angle((0.0, 1.0, 0.0), angle)


@dataclass
class Camera:
    """Thin-lens camera; Camera.new mirrors crates/rt/src/camera.rs:35-57."""
    width: int
    height: int
    viewport_half_height: float
    viewport_half_width: float
    focal_length: float
    center_of_lens: Sequence[float]
    rotation: Sequence[float]
    aperture: float

    @staticmethod
    def new(width, height, vfov, focal_length, center_of_lens, rotation, aperture):
        half_height_factor = f32(np.tan(f32(vfov) / f32(2.0)))
        aspect_ratio = f32(width) / f32(height)
        return Camera(
            width=int(width),
            height=int(height),
            viewport_half_height=float(f32(focal_length) * half_height_factor),
            viewport_half_width=float(f32(focal_length) * half_height_factor * aspect_ratio),
            focal_length=float(f32(focal_length)),
            center_of_lens=tuple(float(x) for x in center_of_lens),
            rotation=tuple(float(x) for x in rotation),
            aperture=float(aperture),
        )

    @staticmethod
    def from_dimensions(width, height):
        """The fixed camera of crates/rt-cli/src/utils/mod.rs:26-45: at the origin, looking down -Z, vfov 70 deg."""
        look_at = np.array([0.0, 0.0, -1.0], dtype=np.float32)
        look_from = np.zeros(3, dtype=np.float32)
        direction = look_at - look_from
        return Camera.new(
            width,
            height,
            f32(math.radians(70.0)),
            f32(np.sqrt(np.dot(direction, direction))),
            look_from,
            look_at_quat(direction, (0.0, 0.0, -1.0)),
            0.0,
        )


@dataclass
class RenderRange:
    """Half-open pixel window, crates/rt-cli/src/utils/mod.rs:47-87."""
    x: range
    y: range

    @staticmethod
    def parse(s: str) -> "RenderRange":
        def one(tok):
            if ".." in tok:
                a, b = tok.split("..", 1)
                return range(int(a), int(b))
            v = int(tok)
            return range(v, v + 1)

        parts = s.split("x")
        if len(parts) != 2:
            raise ValueError("expected `x`x`y`, e.g. 1x4 or 1..7x4..45")
        return RenderRange(one(parts[0]), one(parts[1]))


def parse_spp(s: str) -> range:
    """crates/rt-cli/src/utils/mod.rs:99-109: "7..84" or "32" (= 0..32)."""
    if ".." in s:
        a, b = s.split("..", 1)
        return range(int(a), int(b))
    return range(0, int(s))


def parse_dimensions(s: str):
    """crates/rt-cli/src/utils/mod.rs:163-176."""
    parts = s.split("x")
    if len(parts) < 2:
        raise ValueError("Incorrect format, see help")
    return int(parts[0]), int(parts[1])
