#!/usr/bin/env python3
"""A/B timing of library builds on one box: renders a configuration with each shared library given (paths; "-" = the tree's own)
through the handful of entry points every build has, so that older builds can be compared too.
Usage: python scripts/ab_render.py c2|c2mnee|c5 [spp] -- lib1 lib2 ...      One JSON line per library, best of three renders.
(c2mnee: the Cornell box with use_mnee = 1, 16 spp.)  AB_SHARDS=N in the environment renders shard 0 of N (every N-th row)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_pkg  # noqa: E402

pkg = rtx_pkg.load()
abi = pkg.abi
args = sys.argv[1:]
split = args.index("--")
cfg, libs = args[0], args[split + 1:]
spp = int(args[1]) if split > 1 else 0
if cfg == "c5":
    sd, (W, H, SPP, depth) = pkg.synthetic.stress_scene(10_000_000), (1920, 1080, 64, 8)
else:
    sd, (W, H, SPP, depth) = pkg.scene_loader.load(os.path.join(ROOT, "tests", "golden", "scenes", "cornell_box.json")), (1920, 1080, 16 if cfg == "c2mnee" else 1024, 500)
spp = spp or SPP
abi._share_torch_hip_runtime()
import torch  # noqa: E402
out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
desc = sd.desc()
cam = sd.camera(W, H)
st = sd.settings(width=W, height=H, samples_per_pixel=spp, max_depth=depth, use_nee=1, use_mis=1, seed=1234, mode=abi.MODE_PATHTRACE, precision=abi.PRECISION_F32,
                 use_mnee=1 if cfg == "c2mnee" else 0, shard_index=0, shard_count=int(os.environ.get("AB_SHARDS", "1")))
first = None
for path in libs:
    p = abi.LIB_PATH if path == "-" else os.path.join(ROOT, path)
    lib = C.CDLL(p)
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
    best = None
    for _ in range(3):
        s = abi.Stats()
        assert lib.rtx_render_device(ctx, C.byref(st), C.c_void_p(out.data_ptr()), None, C.byref(s)) == 0, lib.rtx_last_error()
        if best is None or s.seconds_render < best.seconds_render:
            best = s
    digest = float(out.double().sum().item())
    first = digest if first is None else first
    print(json.dumps({"lib": path, "config": cfg, "spp": spp, "shards": int(os.environ.get("AB_SHARDS", "1")), "frame_ms": best.seconds_render * 1e3, "main_kernel_ms": best.seconds_main_kernel * 1e3,
                      "closest_hit_kernel_ms": best.seconds_extend * 1e3, "msamples_per_s": best.samples / best.seconds_render / 1e6,
                      "film_sum_equals_first": digest == first}), flush=True)
    lib.rtx_destroy(ctx)
