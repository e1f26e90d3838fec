"""relMSE of N-spp GPU films against the reference's converged film, for several seeds and both generators."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_pkg; pkg = rtx_pkg.load(); abi = pkg.abi
def rel_mse(x, r):
    x = np.clip(x, 0, 1); r = np.clip(r, 0, 1)
    return float(np.mean((x - r) ** 2 / (r ** 2 + 0.01)))
for name in sys.argv[1:] or ["environment_demo", "cornell_box", "caustics_test"]:
    fx = np.load(os.path.join(ROOT, "tests/golden/converged", name + ".npz"))
    sd = pkg.scene_loader.load(os.path.join(ROOT, "tests/golden/scenes", name + ".json"))
    W, H, N = int(fx["W"]), int(fx["H"]), int(fx["N"])
    with pkg.renderer.HipPathTracer(0) as pt:
        pt.upload_scene(sd); pt.set_camera(sd.camera(W, H))
        for rng in (abi.RNG_PHILOX, abi.RNG_PCG4D):
            vals = []
            for seed in range(1, 65):
                pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=N, max_depth=int(fx["depth"]), seed=seed, rng=rng))
                vals.append(rel_mse(pt.render().astype(np.float64), fx["film"].astype(np.float64)))
            v = np.array(vals)
            print(name, "rng", rng, "relMSE mean %.5f sd %.5f min %.5f max %.5f" % (v.mean(), v.std(), v.min(), v.max()), "floor", float(fx["noise_floor_relmse"]), flush=True)
