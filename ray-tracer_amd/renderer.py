"""Host-side mirror of the reference's renderer-facing interfaces, bound to the HIP library.

`HipPathTracer` has the method names, call order and output convention of the reference's
GPU plug-in ``raytracer::cuda::GPUPathTracer`` (engine/cuda/cuda_api.hpp:72-110):
``set_settings`` -> ``upload_scene`` -> ``set_camera`` -> ``render``.  `render_scene` plays
the role of the PathTrace branch of ``Renderer::render`` (engine/renderer.cpp:263-481) and
`tone_map` / `quantize` the post-process of engine/renderer.cpp:102-261.

Every entry point goes through the C ABI of include/rtx_hip.h; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import abi
from .scene_loader import SceneData, TM_ACES, TM_NONE, TM_REINHARD, TM_REINHARD_EXT, TM_UNCHARTED2

HIT_DTYPE = np.dtype([("hit", "<i4"), ("prim_type", "<i4"), ("prim_index", "<i4"), ("material_id", "<i4"),
                      ("front_face", "<i4"), ("pad_", "<i4"), ("t", "<f8"), ("point", "<f8", 3), ("normal", "<f8", 3),
                      ("u", "<f8"), ("v", "<f8")])


class HipPathTracer:
    def __init__(self, device_id: int = 0, device_ids: list[int] | None = None, lib=None):
        """One GPU (`device_id`, as GPUPathTracer), or with `device_ids` one context over several GPUs of the node
        (rtx_create_multi): same methods, the rows of the film dealt to the devices and gathered on device_ids[0]."""
        self._lib = lib if lib is not None else abi.load_library()   # `lib`: another build of the library (abi.load_library(path)), for tests
        self._ctx = C.c_void_p()
        if device_ids is None:
            abi.check(self._lib, self._lib.rtx_create(C.byref(self._ctx), device_id), "rtx_create")
        else:
            ids = (C.c_int * len(device_ids))(*device_ids)
            abi.check(self._lib, self._lib.rtx_create_multi(C.byref(self._ctx), ids, len(device_ids)), "rtx_create_multi")
        self._settings: abi.Settings | None = None
        self._scene_keepalive = None
        self.last_stats: abi.Stats | None = None

    def close(self):
        if self._ctx:
            self._lib.rtx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- GPUPathTracer surface -------------------------------------------------------------
    def set_settings(self, settings: abi.Settings):
        self._settings = settings

    def upload_scene(self, scene: SceneData):
        desc = scene.desc()
        self._scene_keepalive = (scene, desc)
        abi.check(self._lib, self._lib.rtx_upload_scene(self._ctx, C.byref(desc)), "rtx_upload_scene")

    def set_camera(self, camera: abi.Camera):
        abi.check(self._lib, self._lib.rtx_set_camera(self._ctx, C.byref(camera)), "rtx_set_camera")

    def render(self, output: np.ndarray | None = None) -> np.ndarray:
        """Render with the current settings into `output` (float32, rows_of_shard*W*3); returns
        it shaped [rows, W, 3], row 0 = bottom."""
        st = self._settings
        if st is None:
            raise abi.RtxError("render before set_settings")
        rows = self._lib.rtx_shard_rows(st.height, st.shard_index, max(1, st.shard_count))
        if output is None:
            output = np.empty((rows, st.width, 3), dtype=np.float32)
        assert output.dtype == np.float32 and output.size == rows * st.width * 3 and output.flags.c_contiguous
        stats = abi.Stats()
        abi.check(self._lib, self._lib.rtx_render(self._ctx, C.byref(st), output.ctypes.data_as(C.POINTER(C.c_float)), C.byref(stats)), "rtx_render")
        self.last_stats = stats
        return output.reshape(rows, st.width, 3)

    # -- additions of this backend -----------------------------------------------------------
    def render_device(self, device_ptr: int, stream: int = 0) -> abi.Stats:
        """Render into device memory (e.g. a torch tensor's data_ptr on this device)."""
        st = self._settings
        if st is None:
            raise abi.RtxError("render before set_settings")
        stats = abi.Stats()
        abi.check(self._lib, self._lib.rtx_render_device(self._ctx, C.byref(st), C.c_void_p(device_ptr), C.c_void_p(stream), C.byref(stats)), "rtx_render_device")
        self.last_stats = stats
        return stats

    def render_device_async(self, device_ptr: int, stream: int):
        """rtx_render_device_async: enqueue the render on `stream` and return; `render_wait` collects it."""
        st = self._settings
        if st is None:
            raise abi.RtxError("render before set_settings")
        abi.check(self._lib, self._lib.rtx_render_device_async(self._ctx, C.byref(st), C.c_void_p(device_ptr), C.c_void_p(stream)), "rtx_render_device_async")

    def render_wait(self) -> abi.Stats:
        stats = abi.Stats()
        abi.check(self._lib, self._lib.rtx_render_wait(self._ctx, C.byref(stats)), "rtx_render_wait")
        self.last_stats = stats
        return stats

    def set_gather(self, mode: int):
        """rtx_multi_set_gather: how a multi-device context gathers device-memory films (abi.GATHER_PEER_COPY / GATHER_RCCL)."""
        abi.check(self._lib, self._lib.rtx_multi_set_gather(self._ctx, int(mode)), "rtx_multi_set_gather")

    def render_device_noalloc(self, device_ptr: int, stream: int = 0, want_stats: bool = True) -> abi.Stats | None:
        """rtx_render_device_noalloc: the same render without any device allocation (small films, fused pipeline)."""
        st = self._settings
        if st is None:
            raise abi.RtxError("render before set_settings")
        stats = abi.Stats() if want_stats else None
        abi.check(self._lib, self._lib.rtx_render_device_noalloc(self._ctx, C.byref(st), C.c_void_p(device_ptr), C.c_void_p(stream),
                                                                  C.byref(stats) if want_stats else None), "rtx_render_device_noalloc")
        if want_stats:
            self.last_stats = stats
        return stats

    def postprocess(self, film: np.ndarray, tone_mapper: int = TM_ACES, exposure: float = 1.0) -> np.ndarray:
        """Image::apply_tone_mapping + the quantisation of Image::write_png (engine/renderer.cpp:186-198,
        :232-258) on the GPU: uint8 [H, W, 3], row 0 = TOP, from a float film with row 0 = bottom."""
        film = np.ascontiguousarray(film, dtype=np.float32)
        h, w, ch = film.shape
        assert ch == 3
        out = np.empty((h, w, 3), dtype=np.uint8)
        abi.check(self._lib, self._lib.rtx_postprocess(self._ctx, film.ctypes.data_as(C.POINTER(C.c_float)), w, h, int(tone_mapper), float(exposure),
                                                       out.ctypes.data_as(C.POINTER(C.c_uint8))), "rtx_postprocess")
        return out

    def postprocess_device(self, film_ptr: int, width: int, height: int, out_ptr: int, tone_mapper: int = TM_ACES, exposure: float = 1.0, stream: int = 0):
        """Same on device buffers (e.g. torch tensors' data_ptr on this device)."""
        abi.check(self._lib, self._lib.rtx_postprocess_device(self._ctx, C.c_void_p(film_ptr), width, height, int(tone_mapper), float(exposure),
                                                              C.c_void_p(out_ptr), C.c_void_p(stream)), "rtx_postprocess_device")

    def trace_closest(self, rays8: np.ndarray, fast_precision: int | None = None) -> np.ndarray:
        """Closest hits for rays [n, 8] = (origin, direction, t_min, t_max).  Default: the
        double-precision reference-order kernel; fast_precision selects the production BVH."""
        rays8 = np.ascontiguousarray(rays8, dtype=np.float64).reshape(-1, 8)
        out = np.zeros(rays8.shape[0], dtype=HIT_DTYPE)
        rp = rays8.ctypes.data_as(C.POINTER(C.c_double))
        op = C.cast(out.ctypes.data, C.POINTER(abi.Hit))
        if fast_precision is None:
            abi.check(self._lib, self._lib.rtx_trace_closest(self._ctx, rp, rays8.shape[0], op), "rtx_trace_closest")
        else:
            abi.check(self._lib, self._lib.rtx_trace_closest_fast(self._ctx, rp, rays8.shape[0], fast_precision, op), "rtx_trace_closest_fast")
        return out

    def bvh_info(self) -> abi.BvhInfo:
        info = abi.BvhInfo()
        abi.check(self._lib, self._lib.rtx_get_bvh_info(self._ctx, C.byref(info)), "rtx_get_bvh_info")
        return info


    def reference_bvh(self):
        """The reference-order tree as the library built it (rtx_get_reference_bvh): (boxes [n, 6] float64 = lo, hi;
        nodes [n, 4] int32 = left, right, first, count; prims [m, 2] int32 = type, index)."""
        info = self.bvh_info()
        n, m = int(info.ref_nodes), int(info.ref_prims)
        node_dtype = np.dtype([("box", np.float64, 6), ("ints", np.int32, 4)])
        nodes = np.zeros(n, dtype=node_dtype)
        types, indices = np.zeros(m, dtype=np.int32), np.zeros(m, dtype=np.int32)
        abi.check(self._lib, self._lib.rtx_get_reference_bvh(self._ctx, nodes.ctypes.data, n, types.ctypes.data, indices.ctypes.data, m), "rtx_get_reference_bvh")
        return np.ascontiguousarray(nodes["box"]), np.ascontiguousarray(nodes["ints"]), np.stack([types, indices], axis=1)


def test_hip(device_id: int = 0) -> bool:
    """Counterpart of raytracer::cuda::test_cuda() (engine/cuda/cuda_api.hpp:17)."""
    lib = abi.load_library()
    return lib.rtx_selftest(device_id) == 0


def render_scene(scene: SceneData, device_id: int = 0, **settings_overrides) -> tuple[np.ndarray, abi.Stats]:
    """Load-to-film convenience: the PathTrace branch of Renderer::render on the GPU.
    Returns the linear film [H, W, 3] (row 0 = bottom) and the backend's statistics."""
    st = scene.settings(**settings_overrides)
    with HipPathTracer(device_id) as pt:
        pt.set_settings(st)
        pt.upload_scene(scene)
        pt.set_camera(scene.camera(st.width, st.height))
        film = pt.render()
        return film, pt.last_stats


# ---- post-process (engine/renderer.cpp:102-261) ---------------------------------------------
def tone_map(film: np.ndarray, mapper: int = TM_ACES, exposure: float = 1.0) -> np.ndarray:
    """Image::apply_tone_mapping (engine/renderer.cpp:186-198) in double precision."""
    c = film.astype(np.float64) * exposure
    if mapper == TM_REINHARD:
        return c / (1.0 + c)
    if mapper == TM_REINHARD_EXT:
        wp2 = 4.0 * 4.0
        return c * (1.0 + c / wp2) / (1.0 + c)
    if mapper == TM_ACES:
        a, b, cc, d, e = 2.51, 0.03, 2.43, 0.59, 0.14
        return np.clip((c * (a * c + b)) / (c * (cc * c + d) + e), 0.0, 1.0)
    if mapper == TM_UNCHARTED2:
        def part(x):
            A, B, Cc, D, E, F = 0.15, 0.50, 0.10, 0.20, 0.02, 0.30
            return ((x * (A * x + Cc * B) + D * E) / (x * (A * x + B) + D * F)) - E / F
        return part(c * 2.0) * (1.0 / part(11.2))
    return np.clip(c, 0.0, 1.0)  # ToneMapper::None still clamps (renderer.cpp:175-182)


def write_ppm(path: str, rgb8: np.ndarray) -> None:
    """Image::write_ppm's file format (engine/renderer.cpp:202-230): plain-text P3, one pixel per line,
    top row first.  `rgb8` is uint8 [H, W, 3] with row 0 = top (what `HipPathTracer.postprocess` returns)."""
    h, w, _ = rgb8.shape
    with open(path, "w") as f:
        f.write(f"P3\n{w} {h}\n255\n")
        f.write("".join(f"{r} {g} {b}\n" for r, g, b in rgb8.reshape(-1, 3).tolist()))


def quantize(mapped: np.ndarray) -> np.ndarray:
    """sqrt gamma, 255.999 quantisation and the vertical flip of Image::write_png
    (engine/renderer.cpp:232-258): returns uint8 [H, W, 3] with row 0 = TOP."""
    g = np.sqrt(np.clip(mapped, 0.0, 1.0))
    return (255.999 * g).astype(np.int32).astype(np.uint8)[::-1].copy()
