#!/usr/bin/env python3
"""One GPU listed twice in a multi-device context: two contexts render interleaved halves of the frame (or shard) side by side, so
that the ends of one's kernels overlap with the other's work.  Times shard 0 of N of the 10 M-triangle frame that way and plainly.
    python scripts/twin_contexts.py [N ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_pkg
pkg = rtx_pkg.load(); abi = pkg.abi
import torch
W, H, spp, depth = 1920, 1080, 64, 8
sd = pkg.synthetic.stress_scene(10_000_000)
out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
for ids in ([0], [0, 0], [0, 0, 0]):
    pt = pkg.renderer.HipPathTracer(device_ids=ids) if len(ids) > 1 else pkg.renderer.HipPathTracer(0)
    pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=spp, max_depth=depth, seed=1234))
    pt.upload_scene(sd); pt.set_camera(sd.camera(W, H)); pt.bvh_info()
    for n in [int(v) for v in sys.argv[1:]] or [8, 2, 1]:
        pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=spp, max_depth=depth, use_nee=1, use_mis=1, seed=1234, mode=abi.MODE_PATHTRACE,
                                    precision=abi.PRECISION_F32, shard_index=0, shard_count=n))
        best = None
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            s = pt.render_device(out.data_ptr()); torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print(json.dumps({"contexts_on_gpu0": len(ids), "shards": n, "wall_ms": best * 1e3, "seconds_render_ms": s.seconds_render * 1e3, "film_sum": float(out[: len(range(0, H, n))].double().sum().item())}), flush=True)
    pt.close()
