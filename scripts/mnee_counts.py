#!/usr/bin/env python3
"""Counters of a manifold-NEE render, single precision, from the kernel that batches a wave's solves and from the one that solves at
the vertex (RTX_MNEE_ON_THE_SPOT=1):  python scripts/mnee_counts.py [scene] [W H spp]   (RTX_HIP_LIB picks the library)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_pkg
pkg = rtx_pkg.load(); abi = pkg.abi
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
W, H, spp = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (96, 64, 24)
sd = pkg.scene_loader.load(os.path.join(ROOT, "tests", "golden", "scenes", name + ".json"))
kw = dict(width=W, height=H, samples_per_pixel=spp, max_depth=min(sd.max_depth, 12), seed=5, use_nee=1, use_mis=1, use_mnee=1, precision=abi.PRECISION_F32, pipeline=abi.PIPELINE_FUSED)
films = {}
for tag, env in (("batched", None), ("on_the_spot", "1")):
    os.environ.pop("RTX_MNEE_ON_THE_SPOT", None)
    if env: os.environ["RTX_MNEE_ON_THE_SPOT"] = env
    with pkg.renderer.HipPathTracer(0) as pt:
        pt.upload_scene(sd); pt.set_camera(sd.camera(W, H)); pt.set_settings(sd.settings(**kw))
        films[tag] = pt.render().copy(); st = pt.last_stats
        print(json.dumps({"lib": os.path.basename(abi.LIB_PATH), "kernel": tag, "rays_closest": int(st.rays_closest), "rays_shadow": int(st.rays_shadow), "segments": int(st.segments),
                          "ms": st.seconds_render * 1e3, "film_mean": float(films[tag].mean())}), flush=True)
import numpy as np
d = np.abs(films["batched"].astype(np.float64) - films["on_the_spot"]).max(axis=2)
print(json.dumps({"pixels_differing_1e-4": float((d > 1e-4 * (1 + np.abs(films["on_the_spot"]).max())).mean())}))
