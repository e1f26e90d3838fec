"""Diagnostic: film means and ray counts of both pipelines under each forced intersection mode."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_pkg; pkg = rtx_pkg.load(); abi = pkg.abi
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
depth = int(os.environ.get("DEPTH", "8"))
sd = pkg.scene_loader.load(os.path.join(ROOT, "tests/golden/scenes", name + ".json"))
W, H = 60, 60
with pkg.renderer.HipPathTracer(0) as pt:
    pt.upload_scene(sd); pt.set_camera(sd.camera(W, H))
    for force in os.environ.get("FORCE", "n,0,1").replace("n", "").split(","):
        force = force or None
        if force is None: os.environ.pop("RTX_FORCE_MODE", None)
        else: os.environ["RTX_FORCE_MODE"] = force
        for prec, pipe in ((abi.PRECISION_F64, abi.PIPELINE_FUSED), (abi.PRECISION_F32, abi.PIPELINE_FUSED), (abi.PRECISION_F32, abi.PIPELINE_WAVEFRONT)):
            pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=64, max_depth=depth, seed=77, pipeline=pipe, precision=prec))
            film = pt.render(); s = pt.last_stats
            print(f"force={force} prec={prec} pipe={pipe} mean={film.mean():.6f} closest={s.rays_closest} shadow={s.rays_shadow} seg={s.segments}", flush=True)
