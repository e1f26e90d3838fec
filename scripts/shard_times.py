#!/usr/bin/env python3
"""What one GPU of N spends on its share of a frame: renders shard k of N (rows k, k + N, ...) of the bench configurations on ONE
GPU, for N = 1, 2, 4, 8 and k = 0 and N - 1, and prints the times.  t(1) / (N x t(N)) is the share of linear scaling the render
kernels themselves leave to an N-GPU run (row interleaving evens the load; what a multi-GPU step adds on top - the film gather and
the wait for the slowest rank - is what `bench.py --gpus N` measures on real hardware).

    python scripts/shard_times.py [c2 c3 c5]        one JSON line per (configuration, scene, N)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_pkg  # noqa: E402
from bench import CONFIGS, SCENES  # noqa: E402

pkg = rtx_pkg.load()
abi = pkg.abi
import torch  # noqa: E402

for cfg in (sys.argv[1:] or ["c2", "c3", "c5"]):
    for (name, W, H, spp, depth) in CONFIGS[cfg]:
        sd = pkg.synthetic.stress_scene(10_000_000) if name == "stress10m" else pkg.scene_loader.load(os.path.join(SCENES, name + ".json"))
        pt = pkg.renderer.HipPathTracer(0)
        base = dict(width=W, height=H, samples_per_pixel=spp, max_depth=depth, use_nee=1, use_mis=1, seed=1234, mode=abi.MODE_PATHTRACE, precision=abi.PRECISION_F32)
        pt.set_settings(sd.settings(**base))
        pt.upload_scene(sd)
        pt.set_camera(sd.camera(W, H))
        pt.bvh_info()   # (waits for the reference-order tree of a large scene)
        out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        t1 = None
        for n in (1, 2, 4, 8):
            times = {}
            for k in sorted({0, n - 1}):
                pt.set_settings(sd.settings(shard_index=k, shard_count=n, **base))
                best = None
                for _ in range(3):
                    s = pt.render_device(out.data_ptr())
                    best = s.seconds_render if best is None else min(best, s.seconds_render)
                times[k] = best * 1e3
            worst = max(times.values())
            t1 = worst if n == 1 else t1
            print(json.dumps({"config": cfg, "scene": name, "shards": n, "ms_shard_first": times[0], "ms_shard_last": times[n - 1], "kernel_scaling_efficiency": t1 / (n * worst),
                              "msamples_per_s_times_n": W * H * spp / n / (worst * 1e-3) / 1e6 * n}), flush=True)
        pt.close()
