"""Small array helpers shared by the tests."""
import numpy as np


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same_bits(a, b):
    """Bitwise equality of two f32 arrays, any NaN matching any NaN."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))


def full_frame_xy(width, height):
    ys, xs = np.mgrid[0:height, 0:width]
    return np.stack([xs.ravel(), ys.ravel()], 1).astype(np.uint32)


def pairing_torture_mesh(seed=5, n=24):
    """A wavy n x n grid of quads as indexed triangles, made awkward on purpose for a builder that pairs triangles
    into shared records: mixed diagonals, a third of the triangles with rotated or reversed vertex order, duplicate
    triangles, triangles hanging off an edge that already has two (non-manifold), zero-area triangles.
    Returns (vertices float32 [V, 3], indices uint32 [T, 3])."""
    rng = np.random.default_rng(seed)
    gx, gy = np.meshgrid(np.linspace(-1, 1, n + 1), np.linspace(-1, 1, n + 1), indexing="ij")
    gz = 0.15 * np.sin(3 * gx) * np.cos(2 * gy) + 0.01 * rng.normal(size=gx.shape)
    v = np.stack([gx, gy, gz], -1).reshape(-1, 3).astype(np.float32)
    vid = lambda i, j: i * (n + 1) + j
    tris = []
    for i in range(n):
        for j in range(n):
            a, b, c, d = vid(i, j), vid(i + 1, j), vid(i + 1, j + 1), vid(i, j + 1)
            if (i + j) % 3 == 0:
                tris += [(a, b, d), (b, c, d)]
            else:
                tris += [(a, b, c), (a, c, d)]
    tris = np.array(tris, dtype=np.uint32)
    for k in range(len(tris)):
        r = rng.integers(0, 9)
        if r < 2:
            tris[k] = np.roll(tris[k], int(r) + 1)
        elif r == 2:
            tris[k] = tris[k][::-1]
    extra = [tris[5], tris[5][::-1], tris[40], (vid(3, 3), vid(4, 3), vid(3, 3)), (vid(7, 7), vid(7, 7), vid(7, 7))]
    apex = len(v)
    v = np.vstack([v, np.array([[0.0, 0.0, 0.8], [0.3, -0.2, -0.7]], dtype=np.float32)])
    extra += [(vid(10, 10), vid(11, 10), apex), (vid(11, 10), vid(10, 10), apex + 1)]  # four triangles on one edge
    tris = np.vstack([tris, np.array(extra, dtype=np.uint32)])
    return v, tris


def insert_pairing_torture(scene, material=0):
    """Geometry 0: the torture mesh; 1: an empty mesh; 2: the same points again as another geometry (records must not
    pair across geometries, and every tie goes to geometry 0); 3: a shifted copy; 4: a sphere."""
    v, t = pairing_torture_mesh()
    scene.insert_mesh(material, v, t)
    scene.insert_mesh(material, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32))
    scene.insert_mesh(material, v, t[::-1].copy())
    scene.insert_mesh(material, v + np.array([0.01, 0.0, 0.07], np.float32), t)
    scene.insert_sphere(material, (0.1, 0.2, 0.4), 0.3)


def torture_rays(ray_dtype, n=4000, seed=9):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, dtype=ray_dtype)
    rays["org"] = rng.uniform(-1.2, 1.2, (n, 3)) * np.array([1, 1, 0.2]) + np.array([0, 0, 1.0])
    tgt = rng.uniform(-1, 1, (n, 3)) * np.array([1, 1, 0.2])
    d = tgt - rays["org"]
    rays["dir"] = d / np.linalg.norm(d, axis=1)[:, None]
    rays["org"][n // 2:] *= np.array([1, 1, -1])
    rays["dir"][n // 2:] *= np.array([1, 1, -1])
    rays["tnear"] = 1e-5
    rays["tfar"] = np.inf
    return rays
