#!/usr/bin/env python3
"""Wall time of the PathPhoton photon pass (trace kernel + kd-tree and boxes) for several photon counts, with each library given
("-" = the tree's own).  Usage: python scripts/photon_pass_time.py lib1 lib2 ...     One JSON line per (library, photon count)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_pkg  # noqa: E402

pkg = rtx_pkg.load()
abi = pkg.abi
abi._share_torch_hip_runtime()
import torch  # noqa: E402

sd = pkg.scene_loader.load(os.path.join(ROOT, "tests", "golden", "scenes", "cornell_box.json"))
W, H = 640, 360
out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
desc, cam = sd.desc(), sd.camera(W, H)
for path in sys.argv[1:] or ["-"]:
    lib = C.CDLL(abi.LIB_PATH if path == "-" else os.path.join(ROOT, path))
    lib.rtx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.rtx_upload_scene.argtypes = [C.c_void_p, C.POINTER(abi.SceneDesc)]
    lib.rtx_set_camera.argtypes = [C.c_void_p, C.POINTER(abi.Camera)]
    lib.rtx_render_device.argtypes = [C.c_void_p, C.POINTER(abi.Settings), C.c_void_p, C.c_void_p, C.POINTER(abi.Stats)]
    lib.rtx_destroy.argtypes = [C.c_void_p]
    lib.rtx_last_error.restype = C.c_char_p
    ctx = C.c_void_p()
    assert lib.rtx_create(C.byref(ctx), 0) == 0
    assert lib.rtx_upload_scene(ctx, C.byref(desc)) == 0, lib.rtx_last_error()
    assert lib.rtx_set_camera(ctx, C.byref(cam)) == 0
    for photons in (50_000, 1_000_000, 8_000_000):
        best = None
        for seed in (1, 2, 3):   # a new seed is a new map
            st = sd.settings(width=W, height=H, samples_per_pixel=1, max_depth=8, seed=seed, mode=abi.MODE_PATHPHOTON, precision=abi.PRECISION_F32, caustic_photon_count=photons)
            s = abi.Stats()
            assert lib.rtx_render_device(ctx, C.byref(st), C.c_void_p(out.data_ptr()), None, C.byref(s)) == 0, lib.rtx_last_error()
            best = s.seconds_photon_pass if best is None else min(best, s.seconds_photon_pass)
        print(json.dumps({"lib": path, "caustic_photon_count": photons, "photon_pass_ms": best * 1e3, "film_sum": float(out.double().sum().item())}), flush=True)
    lib.rtx_destroy(ctx)
