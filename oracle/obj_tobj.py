"""OBJ/MTL reading for the CPU oracle (test infrastructure only; the product has its own C++ loader).

Restates what the reference gets out of the `tobj` crate (third-party, pinned at 3.2.5 in the reference's
Cargo.lock, source absent) when called as in crates/rt/src/loader/obj.rs:31-34 with
`GPU_LOAD_OPTIONS` + `single_index` (triangulate, one index stream, points and lines ignored):

* a new model starts at every `o`/`g` line and at every `usemtl` that changes the material while the
  current model already has faces; the last model is always emitted;
* faces are fan-triangulated (a, b, c), (a, c, d), ...; negative indices count back from the vertices
  read so far;
* each model gets its own vertex list, in order of first use, holding only the vertices it references;
* `mtllib` materials are returned in file order with their `Kd`; a missing or unreadable MTL makes the
  material list an error, and then every model falls back to the caller's default material.

and then crates/rt/src/loader/obj.rs:36-122: one DiffuseBxDF per MTL material (Ke is ignored), the transform
applied to every model's positions, one insert_mesh per model.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np


@dataclass
class ObjModel:
    name: str
    positions: np.ndarray  # float32 [n, 3]
    indices: np.ndarray  # uint32 [m, 3]
    material_id: int | None


@dataclass
class ObjMaterial:
    name: str
    diffuse: tuple = (0.0, 0.0, 0.0)
    params: dict = field(default_factory=dict)


def _parse_mtl(path):
    mats = []
    with open(path, "r", errors="replace") as fh:
        for raw in fh:
            line = raw.split("#", 1)[0].strip() if raw.lstrip().startswith("#") else raw.strip()
            if not line:
                continue
            toks = line.split()
            key = toks[0]
            if key == "newmtl":
                mats.append(ObjMaterial(name=line[len("newmtl"):].strip()))
            elif key == "Kd" and mats:
                mats[-1].diffuse = tuple(float(t) for t in toks[1:4])
            elif mats:
                mats[-1].params[key] = toks[1:]
    return mats


def _fix_index(tok, count):
    i = int(tok)
    return count + i if i < 0 else i - 1


def load_obj(path):
    """Returns (models, materials) with materials None when the MTL could not be read."""
    verts = []
    models = []
    materials = None
    mat_map = {}
    name = "unnamed_object"
    mat_id = None
    faces = []  # list of vertex-index tuples (already fan-triangulated)

    def flush():
        remap = {}
        pos = []
        idx = []
        for tri in faces:
            out = []
            for v in tri:
                j = remap.get(v)
                if j is None:
                    j = len(pos)
                    remap[v] = j
                    pos.append(verts[v])
                out.append(j)
            idx.append(out)
        models.append(
            ObjModel(
                name=name,
                positions=np.array(pos, dtype=np.float32).reshape(-1, 3),
                indices=np.array(idx, dtype=np.uint32).reshape(-1, 3),
                material_id=mat_id,
            )
        )

    base = os.path.dirname(os.path.abspath(path))
    with open(path, "r", errors="replace") as fh:
        for raw in fh:
            line = raw.strip()
            if not line or line.startswith("#"):
                continue
            toks = line.split()
            key = toks[0]
            if key == "v":
                verts.append((float(toks[1]), float(toks[2]), float(toks[3])))
            elif key == "f":
                corners = [_fix_index(t.split("/")[0], len(verts)) for t in toks[1:]]
                if len(corners) < 3:
                    raise ValueError("face with fewer than 3 vertices")
                a = corners[0]
                b = corners[1]
                for c in corners[2:]:
                    faces.append((a, b, c))
                    b = c
            elif key in ("o", "g"):
                if faces:
                    flush()
                    faces = []
                name = line[1:].strip() or "unnamed_object"
            elif key == "mtllib":
                try:
                    new = _parse_mtl(os.path.join(base, line[len("mtllib"):].strip()))
                    if materials is None:
                        materials = []
                    for m in new:
                        mat_map[m.name] = len(materials)
                        materials.append(m)
                except OSError:
                    pass
            elif key == "usemtl":
                mname = line[len("usemtl"):].strip()
                if mname:
                    new_mat = mat_map.get(mname)
                    if mat_id != new_mat and faces:
                        flush()
                        faces = []
                    mat_id = new_mat
    flush()
    return models, materials


def apply_transform(positions, translation, scale, rot):
    """crates/rt/src/math/transform.rs:45-52 in f32: glam Quat * v, then scale, then translation."""
    f = np.float32
    p = np.asarray(positions, dtype=np.float32)
    qx, qy, qz, qw = (f(rot[0]), f(rot[1]), f(rot[2]), f(rot[3]))
    b = np.array([qx, qy, qz], dtype=np.float32)
    b2 = f(f(f(qx * qx) + f(qy * qy)) + f(qz * qz))
    k = f(f(qw * qw) - b2)
    vx, vy, vz = p[:, 0], p[:, 1], p[:, 2]
    dot2 = ((vx * b[0] + vy * b[1]) + vz * b[2]) * f(2.0)
    # glam Vec3::cross(b, v)
    cx = b[1] * vz - vy * b[2]
    cy = b[2] * vx - vz * b[0]
    cz = b[0] * vy - vx * b[1]
    w2 = f(qw * f(2.0))
    out = np.empty_like(p)
    for i, (v, c) in enumerate(((vx, cx), (vy, cy), (vz, cz))):
        rotated = (v * k + b[i] * dot2) + c * w2
        out[:, i] = f(scale[i]) * rotated + f(translation[i])
    return out
