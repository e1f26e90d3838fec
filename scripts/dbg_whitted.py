import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_pkg; pkg = rtx_pkg.load(); abi = pkg.abi
name = sys.argv[1]
sd = pkg.scene_loader.load(os.path.join(ROOT, "tests/golden/scenes", name + ".json"))
W, H = 48, 27
with pkg.renderer.HipPathTracer(0) as pt:
    pt.upload_scene(sd); pt.set_camera(sd.camera(W, H))
    out = {}
    for depth in (1, 2, 3):
        for prec in (abi.PRECISION_F64, abi.PRECISION_F32):
            pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=8, max_depth=depth, seed=53, mode=abi.MODE_WHITTED, precision=prec))
            f = pt.render().astype(np.float64).copy(); st = pt.last_stats
            out[prec] = f
            print(name, "depth", depth, "prec", prec, "mean", f.mean(), "closest", st.rays_closest, "shadow", st.rays_shadow, "seg", st.segments)
        d = np.abs(out[0] - out[1]).max(axis=2)
        ys, xs = np.nonzero(d > 1e-3)
        print("  differing", len(ys), [(int(y), int(x), out[1][y, x].round(4).tolist(), out[0][y, x].round(4).tolist()) for y, x in list(zip(ys, xs))[:3]])
