"""Synthetic workloads generated in code (the reference's loader has no mesh import; its
scenes list triangles one JSON object at a time, engine/scene_loader.hpp:239-246)."""
from __future__ import annotations

import numpy as np

from . import abi
from .scene_loader import MODE_PATHTRACE, SceneData, TRIANGLE_DTYPE, make_quad_light, _default_material


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def hash_uniform(index: np.ndarray, stream: int, seed: int) -> np.ndarray:
    """Counter hash -> float64 in [0, 1): value = f(seed, stream, index), no state."""
    with np.errstate(over="ignore"):
        k = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        h = _splitmix64(index.astype(np.uint64) ^ k)
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def triangle_soup(n: int, seed: int = 42, extent: float = 10.0, size: float = 0.05, n_materials: int = 16) -> np.ndarray:
    """n small triangles: centre ~ U[-extent, extent]^3, vertices = centre + size * U[-1, 1]^3,
    material = i mod n_materials.  Returned as a structured array laid out like rtx_triangle."""
    idx = np.arange(n, dtype=np.uint64)
    tri = np.zeros(n, dtype=TRIANGLE_DTYPE)
    centre = np.stack([(hash_uniform(idx, a, seed) * 2.0 - 1.0) * extent for a in range(3)], axis=1)
    for k, name in enumerate(("v0", "v1", "v2")):
        off = np.stack([(hash_uniform(idx, 3 + 3 * k + a, seed) * 2.0 - 1.0) * size for a in range(3)], axis=1)
        tri[name] = centre + off
    tri["material_id"] = (idx % np.uint64(n_materials)).astype(np.int32)
    return tri


def stress_scene(n_triangles: int = 10_000_000, seed: int = 42) -> SceneData:
    """BVH stress scene: a triangle soup inside a Cornell-style box of five planes at +-12
    with one 6x6 quad light under the ceiling; 12 Lambertian, 2 mirror and 2 glass materials;
    camera at (0, 0, 30) looking at the origin, fov 40, 1920x1080, 64 spp, depth 8."""
    sd = SceneData()
    greys = [(0.73, 0.73, 0.73), (0.65, 0.05, 0.05), (0.12, 0.45, 0.15), (0.2, 0.3, 0.7), (0.7, 0.6, 0.2), (0.5, 0.5, 0.5),
             (0.9, 0.9, 0.9), (0.3, 0.3, 0.3), (0.6, 0.2, 0.6), (0.2, 0.6, 0.6), (0.8, 0.4, 0.1), (0.4, 0.8, 0.4)]
    for c in greys:
        m = _default_material()
        m.albedo[:] = c
        sd.materials.append(m)
    for _ in range(2):
        m = _default_material()
        m.type = abi.MAT_METAL
        m.albedo[:] = (0.95, 0.95, 0.95)
        sd.materials.append(m)
    for _ in range(2):
        m = _default_material()
        m.type = abi.MAT_DIELECTRIC
        m.albedo[:] = (1.0, 1.0, 1.0)
        m.refraction_index = 1.5
        sd.materials.append(m)
    sd.material_names = [f"m{i}" for i in range(len(sd.materials))]
    for point, normal, mat in (((0, -12, 0), (0, 1, 0), 0), ((0, 12, 0), (0, -1, 0), 0), ((0, 0, -12), (0, 0, 1), 0),
                               ((-12, 0, 0), (1, 0, 0), 1), ((12, 0, 0), (-1, 0, 0), 2)):
        p = abi.Plane()
        p.point[:] = tuple(float(v) for v in point)
        p.normal[:] = tuple(float(v) for v in normal)
        p.material_id = mat
        sd.planes.append(p)
    sd.quad_lights.append(make_quad_light((-3.0, 11.99, -3.0), (6.0, 0.0, 0.0), (0.0, 0.0, 6.0), (15.0, 15.0, 15.0)))
    sd.triangle_array = triangle_soup(n_triangles, seed)
    sd.camera_position, sd.camera_target, sd.camera_up, sd.camera_fov = (0.0, 0.0, 30.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 40.0
    sd.width, sd.height, sd.samples, sd.max_depth = 1920, 1080, 64, 8
    sd.mode = MODE_PATHTRACE
    return sd


def pixel_centre_rays(camera: abi.Camera, width: int, height: int, t_min: float = 0.001, t_max: float = 1e9) -> np.ndarray:
    """One ray per pixel through its centre, [height*width, 8] = (origin, unnormalised direction, t_min, t_max), built as
    Camera::get_ray builds it (engine/scene.hpp:99-108) for s = (x + 0.5) / W, t = (y + 0.5) / H."""
    o = np.array(camera.origin[:])
    llc, hz, vt = np.array(camera.lower_left_corner[:]), np.array(camera.horizontal[:]), np.array(camera.vertical[:])
    xs, ys = np.meshgrid(np.arange(width), np.arange(height))
    s = ((xs + 0.5) / width).ravel()
    t = ((ys + 0.5) / height).ravel()
    d = ((llc[None] + s[:, None] * hz[None]) + t[:, None] * vt[None]) - o[None]
    n = width * height
    return np.concatenate([np.tile(o, (n, 1)), d, np.full((n, 1), t_min), np.full((n, 1), t_max)], axis=1)


def stress_rays(n: int, seed: int = 4242, half: float = 11.9) -> np.ndarray:
    """n incoherent rays for the stress scene, a pure function of (n, seed) computed with + - * only (no library function
    whose rounding could differ between hosts): origins uniform in [-half, half]^3, unnormalised directions uniform in a cube
    scaled by a factor in [0.5, 2) (the reference never normalises a ray); tenths of the set have a zero x / y / -0.0 z
    component, are axis-parallel, or are short (t_max 3); t_min = 0.001."""
    idx = np.arange(n, dtype=np.uint64)
    u = [hash_uniform(idx, k, seed) for k in range(7)]
    org = np.stack([(u[0] * 2.0 - 1.0) * half, (u[1] * 2.0 - 1.0) * half, (u[2] * 2.0 - 1.0) * half], axis=1)
    d = np.stack([u[3] * 2.0 - 1.0, u[4] * 2.0 - 1.0, u[5] * 2.0 - 1.0], axis=1) * (0.5 + 1.5 * u[6])[:, None]
    k = n // 10
    d[0 * k:1 * k, 0] = 0.0
    d[1 * k:2 * k, 1] = 0.0
    d[2 * k:3 * k, 2] = -0.0
    d[3 * k:4 * k, 1:] = 0.0
    d[3 * k:4 * k, 0] = np.where(d[3 * k:4 * k, 0] == 0.0, 1.0, d[3 * k:4 * k, 0])
    t_max = np.full(n, 1e9)
    t_max[4 * k:5 * k] = 3.0
    return np.concatenate([org, d, np.full((n, 1), 0.001), t_max[:, None]], axis=1)


def hit_table_digest(hits: np.ndarray) -> str:
    """SHA-256 over the canonical closest-hit table of a ray set: per ray (hit, primitive type, primitive index, bits of t),
    misses as (0, -1, -1, 0).  `hits` is a structured array with the fields of rtx_hit."""
    import hashlib
    rec = np.zeros(hits.shape[0], dtype=np.dtype([("hit", "<i4"), ("type", "<i4"), ("index", "<i4"), ("pad", "<i4"), ("t", "<u8")]))
    h = hits["hit"] != 0
    rec["hit"] = h
    rec["type"] = np.where(h, hits["prim_type"], -1)
    rec["index"] = np.where(h, hits["prim_index"], -1)
    rec["t"] = np.where(h, np.ascontiguousarray(hits["t"], dtype=np.float64).view(np.uint64), 0)
    return hashlib.sha256(rec.tobytes()).hexdigest()
