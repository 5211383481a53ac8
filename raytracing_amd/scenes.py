"""The reference's example scenes as data, written against the SceneT surface.

Each `insert_into(scene)` takes any object offering insert_material / insert_light / insert_mesh / insert_sphere /
load_obj with the meaning of crates/rt/src/scene/mod.rs:8-26 and crates/rt/src/loader/obj.rs:15-22, and issues the
same calls, in the same order, as the reference scene it restates:

  CornellBoxScene       crates/rt/src/scene/examples/cornell.rs:13-42
  StandfordBunnyScene   crates/rt/src/scene/examples/standford_bunny.rs:12-29
  DragonScene           crates/rt/src/scene/examples/dragon.rs:16-47
  SpheresScene          crates/rt/src/scene/examples/spheres.rs:10-66
  DebugScene            crates/rt/src/scene/examples/debug.rs:6-16

`obj/dragon.obj` is not distributed with the reference checkout.  When no such file is found, DragonScene inserts a
deterministic stand-in of comparable size instead: the Stanford bunny subdivided five times (4968 * 4^5 =
5,087,232 triangles) with a hashed fractal displacement, scaled so that it covers the same share of the frame as
the dragon in the reference's own render of the scene (see standin.py).
"""
from __future__ import annotations

import math
import os

import numpy as np

from .types import (DielectricBxDF, DiffuseBxDF, EmitBxDF, LightDescriptor, MaterialDescriptor, QUAT_IDENTITY,
                    Transform, quat_from_axis_angle)

_REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def obj_dir():
    return os.environ.get("RTMI_OBJ_DIR", os.path.join(_REPO_ROOT, "assets", "obj"))


def obj_path(name):
    return os.path.join(obj_dir(), name)


class CornellBoxScene:
    @staticmethod
    def insert_into(scene):
        default_material2 = scene.insert_material(
            MaterialDescriptor("Goosh - Default 2", DiffuseBxDF(albedo=(5.5, 0.8, 0.9))))
        scene.load_obj(obj_path("cornell_box.obj"),
                       Transform(translation=(0.0, -0.5, -0.5), scale=(0.5, 0.5, 0.5), rot=QUAT_IDENTITY),
                       default_material2)
        scene.insert_light(LightDescriptor(None, (0.0, 0.4, -0.4)))
        light = scene.insert_material(MaterialDescriptor("light!", EmitBxDF(le=(5.0, 5.0, 5.0))))
        scene.insert_sphere(light, (0.0, 0.0, 5.0), 3.0)


class StandfordBunnyScene:
    @staticmethod
    def insert_into(scene):
        default_material = scene.insert_material(
            MaterialDescriptor("Goosh - Default", DiffuseBxDF(albedo=(1.0, 1.0, 0.5))))
        scene.load_obj(obj_path("standford_bunny.obj"),
                       Transform(translation=(0.2, -0.3, -0.5), scale=(4.0, 4.0, 4.0), rot=QUAT_IDENTITY),
                       default_material)


class DragonScene:
    """Rough-glass dragon + diffuse sphere.  `diffuse=True` swaps the glass for a grey diffuse material (the
    "multi-bounce diffuse" variant one benchmark configuration names); `levels` sizes the stand-in mesh."""

    @staticmethod
    def insert_into(scene, diffuse=False, levels=5, extent=None):
        if diffuse:
            body = scene.insert_material(MaterialDescriptor(None, DiffuseBxDF(albedo=(0.8, 0.8, 0.8))))
        else:
            body = scene.insert_material(MaterialDescriptor(None, DielectricBxDF(ior=1.5, roughness=0.2)))
        transform = Transform(
            translation=(0.0, 0.0, -1.0),
            scale=(0.01, 0.01, 0.01),
            rot=quat_from_axis_angle((0.0, 1.0, 0.0), np.float32(1.1) * np.float32(math.pi)),
        )
        path = obj_path("dragon.obj")
        if os.path.exists(path):
            scene.load_obj(path, transform, body)
            scene.dragon_mesh = "obj/dragon.obj"
        else:
            from .standin import dragon_standin
            from .standin import STANDIN_EXTENT
            extent = STANDIN_EXTENT if extent is None else float(extent)
            positions, indices = dragon_standin(obj_path("standford_bunny.obj"), levels=levels, extent=extent)
            scene.insert_mesh(body, transform.apply_points(positions), indices)
            scene.dragon_mesh = "stand-in (subdivided bunny, %d triangles, height %g units: the reference dragon's " \
                                "33.8 %% frame coverage)" % (indices.shape[0], extent)
        scene.insert_light(LightDescriptor(None, (10.2, 80.0, 75.0)))
        ball = scene.insert_material(MaterialDescriptor(None, DiffuseBxDF(albedo=(0.2, 0.1, 0.5))))
        scene.insert_sphere(ball, (-0.7, -0.2, -1.9), 0.8)


class SpheresScene:
    @staticmethod
    def insert_into(scene):
        diffuse = scene.insert_material(MaterialDescriptor(None, DiffuseBxDF(albedo=(0.2, 0.9, 0.7))))
        diffuse_blue = scene.insert_material(MaterialDescriptor(None, DiffuseBxDF(albedo=(0.2, 0.4, 0.8))))
        glass = scene.insert_material(MaterialDescriptor(None, DielectricBxDF(ior=1.5, roughness=0.01)))
        scene.insert_sphere(diffuse, (-0.6, 0.05, -1.0), 0.3)
        scene.insert_sphere(diffuse_blue, (-0.3, -0.05, 1.0), 0.2)
        scene.insert_sphere(glass, (0.0, 0.0, -0.3), 0.15)
        scene.insert_light(LightDescriptor(None, (0.0, 0.0, -0.5)))
        scene.insert_light(LightDescriptor(None, (0.4, -0.0, -0.6)))
        scene.insert_light(LightDescriptor(None, (-0.1, -0.1, 0.6)))


class DebugScene:
    @staticmethod
    def insert_into(scene):
        default_material = scene.insert_material(
            MaterialDescriptor("Goosh - Default", DiffuseBxDF(albedo=(1.0, 1.0, 0.0))))
        scene.insert_sphere(default_material, (0.0, 0.0, -1.0), 0.3)
        scene.insert_sphere(default_material, (0.5, 0.0, -1.0), 0.3)


# crates/rt-cli/src/utils/mod.rs:111-131
AVAILABLE_SCENES = {
    "bunny": StandfordBunnyScene,
    "cornell-box": CornellBoxScene,
    "spheres": SpheresScene,
    "debug": DebugScene,
    "dragon": DragonScene,
}
