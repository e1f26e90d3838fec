#!/usr/bin/env python3
"""Which pipeline renders a triangle soup of n triangles faster: the fused tree-walking kernel or the queue pipeline?
    python scripts/pipeline_crossover.py [n ...]      one JSON line per (n, pipeline); 1920x1080, 16 spp, depth 8"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_pkg
pkg = rtx_pkg.load(); abi = pkg.abi
import torch
W, H, spp, depth = 1920, 1080, 16, 8
out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
for n in [int(v) for v in sys.argv[1:]] or [100, 1000, 10_000, 100_000, 1_000_000]:
    sd = pkg.synthetic.stress_scene(n)
    with pkg.renderer.HipPathTracer(0) as pt:
        pt.upload_scene(sd); pt.set_camera(sd.camera(W, H)); pt.bvh_info()
        for name, pipe in (("fused", abi.PIPELINE_FUSED), ("queues", abi.PIPELINE_WAVEFRONT)):
            pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=spp, max_depth=depth, use_nee=1, use_mis=1, seed=1234, mode=abi.MODE_PATHTRACE,
                                        precision=abi.PRECISION_F32, pipeline=pipe))
            best = min(pt.render_device(out.data_ptr()).seconds_render for _ in range(3))
            print(json.dumps({"triangles": n, "pipeline": name, "ms": best * 1e3, "msamples_per_s": W * H * spp / best / 1e6}), flush=True)
