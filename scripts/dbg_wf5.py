"""Diagnostic: first depth at which the wavefront film leaves the fused film (forced mode via RTX_FORCE_MODE)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_pkg; pkg = rtx_pkg.load(); abi = pkg.abi
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
sd = pkg.scene_loader.load(os.path.join(ROOT, "tests/golden/scenes", name + ".json"))
W, H = 60, 60
with pkg.renderer.HipPathTracer(0) as pt:
    pt.upload_scene(sd); pt.set_camera(sd.camera(W, H))
    for nee, mis in ((1, 1), (0, 0)):
        for depth in (1, 2, 3, 4, 5):
            films = []
            for pipe in (abi.PIPELINE_FUSED, abi.PIPELINE_WAVEFRONT):
                pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=4, max_depth=depth, seed=77, pipeline=pipe, use_nee=nee, use_mis=mis))
                films.append(pt.render().astype(np.float64).copy()); s = pt.last_stats
            d = np.abs(films[0] - films[1]).max(axis=2)
            ys, xs = np.nonzero(d > 1e-4)
            print(f"nee={nee} depth={depth} differing={len(ys)} closest={s.rays_closest}", [(int(y), int(x), films[0][y, x].round(4).tolist(), films[1][y, x].round(4).tolist()) for y, x in list(zip(ys, xs))[:4]], flush=True)
