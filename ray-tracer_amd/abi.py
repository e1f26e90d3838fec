"""ctypes mirror of include/rtx_hip.h and loader of the HIP shared library.

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is no
CPU fallback: if the shared object is missing ``load_library`` raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "librtx_hip.so")

d3 = C.c_double * 3


class Sphere(C.Structure):
    _fields_ = [("center", d3), ("radius", C.c_double), ("material_id", C.c_int32), ("pad_", C.c_int32)]


class Plane(C.Structure):
    _fields_ = [("point", d3), ("normal", d3), ("material_id", C.c_int32), ("pad_", C.c_int32)]


class Triangle(C.Structure):
    _fields_ = [("v0", d3), ("v1", d3), ("v2", d3), ("material_id", C.c_int32), ("pad_", C.c_int32)]


class Box(C.Structure):
    _fields_ = [("box_min", d3), ("box_max", d3), ("material_id", C.c_int32), ("pad_", C.c_int32)]


class PointLight(C.Structure):
    _fields_ = [("position", d3), ("intensity", d3)]


class QuadLight(C.Structure):
    _fields_ = [("corner", d3), ("edge_u", d3), ("edge_v", d3), ("normal", d3), ("emission", d3), ("area", C.c_double)]


class DiskLight(C.Structure):
    _fields_ = [("center", d3), ("normal", d3), ("radius", C.c_double), ("emission", d3), ("area", C.c_double),
                ("tangent", d3), ("bitangent", d3)]


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("has_texture", C.c_int32), ("albedo", d3), ("emission", d3),
                ("fuzz", C.c_double), ("refraction_index", C.c_double), ("tex_type", C.c_int32),
                ("tex_image", C.c_int32), ("tex_color1", d3), ("tex_color2", d3), ("tex_scale", C.c_double),
                ("has_normal_map", C.c_int32), ("normal_image", C.c_int32), ("normal_strength", C.c_double)]


class Image(C.Structure):
    _fields_ = [("rgb", C.POINTER(C.c_uint8)), ("width", C.c_int32), ("height", C.c_int32)]


class Environment(C.Structure):
    _fields_ = [("rgb", C.POINTER(C.c_float)), ("width", C.c_int32), ("height", C.c_int32),
                ("intensity", C.c_double), ("rotation", C.c_double)]


class SceneDesc(C.Structure):
    _fields_ = [("spheres", C.POINTER(Sphere)), ("n_spheres", C.c_size_t),
                ("planes", C.POINTER(Plane)), ("n_planes", C.c_size_t),
                ("triangles", C.POINTER(Triangle)), ("n_triangles", C.c_size_t),
                ("boxes", C.POINTER(Box)), ("n_boxes", C.c_size_t),
                ("point_lights", C.POINTER(PointLight)), ("n_point_lights", C.c_size_t),
                ("quad_lights", C.POINTER(QuadLight)), ("n_quad_lights", C.c_size_t),
                ("disk_lights", C.POINTER(DiskLight)), ("n_disk_lights", C.c_size_t),
                ("materials", C.POINTER(Material)), ("n_materials", C.c_size_t),
                ("images", C.POINTER(Image)), ("n_images", C.c_size_t),
                ("environment", C.POINTER(Environment))]


class Camera(C.Structure):
    _fields_ = [("origin", d3), ("lower_left_corner", d3), ("horizontal", d3), ("vertical", d3)]


class Settings(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("samples_per_pixel", C.c_int32),
                ("max_depth", C.c_int32), ("use_nee", C.c_int32), ("use_mis", C.c_int32), ("use_mnee", C.c_int32),
                ("use_background_gradient", C.c_int32), ("background_color", d3), ("background_top", d3),
                ("background_bottom", d3), ("seed", C.c_uint32), ("precision", C.c_int32), ("pipeline", C.c_int32),
                ("shard_index", C.c_int32), ("shard_count", C.c_int32), ("mode", C.c_int32), ("rng", C.c_int32), ("clamp_max", C.c_double),
                ("caustic_photon_count", C.c_uint64), ("photon_gather_count", C.c_int32), ("caustic_gather_radius", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("segments", C.c_uint64), ("seconds_render", C.c_double), ("seconds_main_kernel", C.c_double), ("seconds_total", C.c_double),
                ("kernel_launches", C.c_uint32), ("pipeline_used", C.c_uint32),
                ("precision_used", C.c_uint32), ("flags", C.c_uint32), ("seconds_photon_pass", C.c_double),
                ("seconds_extend", C.c_double), ("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64)]


class Hit(C.Structure):
    _fields_ = [("hit", C.c_int32), ("prim_type", C.c_int32), ("prim_index", C.c_int32), ("material_id", C.c_int32),
                ("front_face", C.c_int32), ("pad_", C.c_int32), ("t", C.c_double), ("point", d3), ("normal", d3),
                ("u", C.c_double), ("v", C.c_double)]


class BvhInfo(C.Structure):
    _fields_ = [("ref_nodes", C.c_uint64), ("ref_prims", C.c_uint64), ("wide_nodes", C.c_uint64),
                ("wide_bytes", C.c_uint64), ("build_seconds", C.c_double), ("leaf_bytes", C.c_uint64), ("wide_leaves", C.c_uint64),
                ("wide_levels", C.c_uint64), ("wide_build_seconds", C.c_double)]


assert C.sizeof(Hit) == 96

PRECISION_F32, PRECISION_F64 = 0, 1
PIPELINE_AUTO, PIPELINE_FUSED, PIPELINE_WAVEFRONT = 0, 1, 2
MODE_WHITTED, MODE_PATHTRACE, MODE_BDPT, MODE_PATHPHOTON = 0, 1, 2, 6
RNG_AUTO, RNG_PHILOX, RNG_PCG4D = 0, 1, 2
GATHER_PEER_COPY, GATHER_RCCL = 0, 1
STAT_BEFORE_TREE, STAT_TIES_REPAIRED, STAT_RENDERED_TWICE, STAT_WALKS_SIDE_BY_SIDE = 1, 2, 4, 8
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC = 0, 1, 2
TEX_SOLID, TEX_CHECKER, TEX_IMAGE = 0, 1, 2

# Every symbol include/rtx_hip.h declares, with its ctypes signature.
EXPORTS = {
    "rtx_selftest": (C.c_int, [C.c_int]),
    "rtx_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "rtx_create_multi": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int]),
    "rtx_multi_shard": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rtx_destroy": (C.c_int, [C.c_void_p]),
    "rtx_upload_scene": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "rtx_set_camera": (C.c_int, [C.c_void_p, C.POINTER(Camera)]),
    "rtx_render": (C.c_int, [C.c_void_p, C.POINTER(Settings), C.POINTER(C.c_float), C.POINTER(Stats)]),
    "rtx_render_device": (C.c_int, [C.c_void_p, C.POINTER(Settings), C.c_void_p, C.c_void_p, C.POINTER(Stats)]),
    "rtx_render_device_async": (C.c_int, [C.c_void_p, C.POINTER(Settings), C.c_void_p, C.c_void_p]),
    "rtx_render_wait": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "rtx_multi_set_gather": (C.c_int, [C.c_void_p, C.c_int]),
    "rtx_render_device_noalloc": (C.c_int, [C.c_void_p, C.POINTER(Settings), C.c_void_p, C.c_void_p, C.POINTER(Stats)]),
    "rtx_postprocess": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_uint8)]),
    "rtx_postprocess_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]),
    "rtx_trace_closest": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.POINTER(Hit)]),
    "rtx_trace_closest_fast": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.c_int, C.POINTER(Hit)]),
    "rtx_shard_rows": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "rtx_get_bvh_info": (C.c_int, [C.c_void_p, C.POINTER(BvhInfo)]),
    "rtx_get_reference_bvh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtx_last_error": (C.c_char_p, []),
}

_lib = None


def _share_torch_hip_runtime():
    """Keep ONE HIP runtime in the process when torch is installed.

    torch's ROCm wheels bundle their own runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7) and link
    against it by FILE name, which the dynamic loader does not match with a runtime already loaded under its
    SONAME from /opt/rocm/lib.  With this library loaded before `import torch`, the process therefore ended up
    with two HIP + HSA runtimes, and the second one cannot initialise the device that the first one holds: torch
    then reports "No HIP GPUs are available".  Loading torch's copy first makes librtx_hip.so (which asks for the
    SONAME) bind to it, whichever of the two modules is imported first.  torch itself is not imported here."""
    if os.environ.get("RTX_KEEP_SYSTEM_HIP_RUNTIME"):
        return
    try:
        with open("/proc/self/maps") as f:
            if "libamdhip64" in f.read():
                return   # a runtime is already in the process: the loader will reuse it by SONAME
    except OSError:
        pass
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    for d in spec.submodule_search_locations:
        cand = os.path.join(d, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            return


def load_library(path: str | None = None):
    """dlopen librtx_hip.so and attach signatures.  Raises if it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("RTX_HIP_LIB") or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    _share_torch_hip_runtime()
    lib = C.CDLL(p)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


class RtxError(RuntimeError):
    pass


def check(lib, code: int, what: str):
    if code != 0:
        msg = lib.rtx_last_error()
        raise RtxError(f"{what} failed ({code}): {msg.decode() if msg else ''}")
