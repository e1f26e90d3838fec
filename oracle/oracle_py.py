"""TEST INFRASTRUCTURE ONLY — ctypes binding of oracle/libpt_oracle.so (the CPU restatement).

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

import rtx_pkg

_pkg = rtx_pkg.load()
abi = _pkg.abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "libpt_oracle.so")
RNG_MT19937, RNG_COUNTER, RNG_COUNTER_PCG = 0, 1, 2
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.orc_create.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(C.c_void_p)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_destroy.restype = None
        L.orc_bvh_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.orc_bvh_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.Settings), C.c_int, C.c_int,
                                 C.c_void_p, C.POINTER(abi.Stats)]
        L.orc_kat.argtypes = [C.c_void_p, C.c_uint, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_philox4x32_10.restype = None
        _lib = L
    return _lib


HIT_DTYPE = np.dtype([("hit", "<i4"), ("prim_type", "<i4"), ("prim_index", "<i4"), ("material_id", "<i4"),
                      ("front_face", "<i4"), ("pad_", "<i4"), ("t", "<f8"), ("point", "<f8", 3), ("normal", "<f8", 3),
                      ("u", "<f8"), ("v", "<f8")])
assert HIT_DTYPE.itemsize == 96


class OracleScene:
    def __init__(self, scene_data):
        self._sd = scene_data
        self._desc = scene_data.desc()
        self._h = C.c_void_p()
        rc = lib().orc_create(C.byref(self._desc), C.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"orc_create failed: {rc}")

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bvh(self):
        nn, npr = C.c_int64(), C.c_int64()
        lib().orc_bvh_counts(self._h, C.byref(nn), C.byref(npr))
        boxes = np.zeros((nn.value, 6), dtype=np.float64)
        nodes = np.zeros((nn.value, 4), dtype=np.int32)
        prims = np.zeros((npr.value, 2), dtype=np.int32)
        lib().orc_bvh_dump(self._h, boxes.ctypes.data, nodes.ctypes.data, prims.ctypes.data)
        return boxes, nodes, prims

    def trace_closest(self, rays8: np.ndarray, threads: int = 8) -> np.ndarray:
        rays8 = np.ascontiguousarray(rays8, dtype=np.float64).reshape(-1, 8)
        out = np.zeros(rays8.shape[0], dtype=HIT_DTYPE)
        rc = lib().orc_trace_closest(self._h, rays8.ctypes.data, rays8.shape[0], out.ctypes.data, threads)
        assert rc == 0
        return out

    def render(self, camera, settings, rng_mode=RNG_COUNTER, threads=8):
        film = np.zeros((settings.height, settings.width, 3), dtype=np.float64)
        st = abi.Stats()
        rc = lib().orc_render(self._h, C.byref(camera), C.byref(settings), rng_mode, threads, film.ctypes.data, C.byref(st))
        if rc != 0:
            raise RuntimeError(f"orc_render failed: {rc}")
        return film, st

    def kat(self, seed: int, n: int) -> np.ndarray:
        cap = n * (64 + 3 * len(self._sd.materials)) + 16
        out = np.zeros(cap, dtype=np.float64)
        w = C.c_size_t()
        rc = lib().orc_kat(self._h, seed, n, out.ctypes.data, cap, C.byref(w))
        assert rc == 0
        return out[: w.value].copy()


def postprocess(film, tone_mapper, exposure):
    """uint8 [H, W, 3], row 0 = top (orc_postprocess)."""
    f = np.ascontiguousarray(film, dtype=np.float32)
    H, W, _ = f.shape
    out = np.zeros((H, W, 3), dtype=np.uint8)
    L = lib()
    L.orc_postprocess.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
    L.orc_postprocess.restype = C.c_int
    rc = L.orc_postprocess(f.ctypes.data, W, H, int(tone_mapper), float(exposure), out.ctypes.data)
    assert rc == 0
    return out


def pcg4d(pixel, sample, block, seed):
    o = np.zeros(4, dtype=np.uint32)
    L = lib()
    L.orc_pcg4d.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.orc_pcg4d.restype = None
    L.orc_pcg4d(int(pixel), int(sample), int(block), int(seed), o.ctypes.data)
    return o


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(c.ctypes.data, k.ctypes.data, o.ctypes.data)
    return o


def set_photon_aim_seed(seed):
    """PathPhoton with RNG_MT19937: the seed of the reference's thread-id-seeded aim generator in the run to reproduce."""
    L = lib()
    L.orc_set_photon_aim_seed.argtypes = [C.c_ulonglong]
    L.orc_set_photon_aim_seed.restype = None
    L.orc_set_photon_aim_seed(int(seed))
