#!/usr/bin/env python3
"""The 10 M-triangle frame (1920x1080, 64 spp, depth 8, queue pipeline) once per value of RTX_ENTRY_LEVELS: how far above its leaf
the entry point of a ray that leaves a primitive lies (0 = every walk starts at the root).  One JSON line per value.
Usage: python scripts/c5_entry_sweep.py [levels ...]   (default 0 1 2 3 4 5)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_pkg  # noqa: E402

pkg = rtx_pkg.load()
abi = pkg.abi
levels = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3, 4, 5]
spp = int(os.environ.get("SWEEP_SPP", "64"))
sd = pkg.synthetic.stress_scene(10_000_000)
W, H = 1920, 1080
import torch  # noqa: E402
out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
first = None
for lv in levels:
    os.environ["RTX_ENTRY_LEVELS"] = str(lv)
    with pkg.renderer.HipPathTracer(0) as pt:
        pt.upload_scene(sd)
        pt.set_camera(sd.camera(W, H))
        pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=spp, max_depth=8, use_nee=1, use_mis=1, seed=1234, mode=abi.MODE_PATHTRACE, precision=abi.PRECISION_F32))
        best = None
        for _ in range(2):
            s = pt.render_device(out.data_ptr())
            best = s if best is None or s.seconds_render < best.seconds_render else best
        digest = float(out.double().sum().item())
        first = digest if first is None else first
        print(json.dumps({"entry_levels": lv, "frame_ms": best.seconds_render * 1e3, "closest_hit_kernel_ms": best.seconds_extend * 1e3,
                          "grays_per_s_closest_hit": best.rays_closest / best.seconds_extend / 1e9, "msamples_per_s": best.samples / best.seconds_render / 1e6,
                          "node_records_per_closest_ray": best.nodes_visited / best.rays_closest, "leaf_records_per_closest_ray": best.prims_tested / best.rays_closest,
                          "node_records_per_shadow_ray": best.nodes_visited_shadow / max(1, best.rays_shadow), "leaf_records_per_shadow_ray": best.prims_tested_shadow / max(1, best.rays_shadow),
                          "rays_closest": best.rays_closest, "rays_shadow": best.rays_shadow, "film_sum_equals_first": digest == first}), flush=True)
