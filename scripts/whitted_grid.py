#!/usr/bin/env python3
"""Whitted renders of a scene that walks a tree (earth_demo as it is in its file, kitchen_sink) under several grid sizes:
RTX_ITEMS_PER_LANE_MODES=<n> gives every lane about n work items instead of the resident grid.  One JSON line per (scene, n)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_pkg
pkg = rtx_pkg.load(); abi = pkg.abi
import torch
W, H, spp = 3840, 2160, 64
out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
for name in ("earth_demo", "kitchen_sink"):
    sd = pkg.scene_loader.load(os.path.join(ROOT, "tests", "golden", "scenes", name + ".json"))
    with pkg.renderer.HipPathTracer(0) as pt:
        pt.upload_scene(sd); pt.set_camera(sd.camera(W, H))
        pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=spp, max_depth=20, seed=3, mode=abi.MODE_WHITTED, precision=abi.PRECISION_F32))
        for v in ("", "24", "16", "12", "8", "6"):
            os.environ.pop("RTX_ITEMS_PER_LANE_MODES", None)
            if v: os.environ["RTX_ITEMS_PER_LANE_MODES"] = v
            best = min(pt.render_device(out.data_ptr()).seconds_render for _ in range(3))
            print(json.dumps({"scene": name, "items_per_lane": v or "resident grid", "ms": best * 1e3, "prims": int(pt.bvh_info().n_prims) if hasattr(pt.bvh_info(), "n_prims") else None}), flush=True)
