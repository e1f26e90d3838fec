import os, sys, numpy as np, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_pkg; pkg = rtx_pkg.load(); abi = pkg.abi
n = int(sys.argv[1])
sd = pkg.synthetic.stress_scene(n)
rng = np.random.default_rng(1)
m = 1 << 20
ro = rng.uniform(-11, 11, (m, 3)); rd = rng.normal(size=(m, 3)); rd /= np.linalg.norm(rd, axis=1, keepdims=True)
rays = np.concatenate([ro, rd, np.full((m, 1), 0.001), np.full((m, 1), 1e9)], axis=1)
cam = sd.camera(1920, 1080)
o = np.array(cam.origin[:]); llc = np.array(cam.lower_left_corner[:]); hz = np.array(cam.horizontal[:]); vt = np.array(cam.vertical[:])
xs, ys = np.meshgrid(np.arange(1024), np.arange(1024)); s = ((xs + .5) / 1024).ravel(); t = ((ys + .5) / 1024).ravel()
d = (llc[None] + s[:, None] * hz[None]) + t[:, None] * vt[None] - o[None]
prim = np.concatenate([np.tile(o, (m, 1)), d, np.full((m, 1), 0.001), np.full((m, 1), 1e9)], axis=1)
with pkg.renderer.HipPathTracer(0) as pt:
    pt.upload_scene(sd)
    for tag, r in (("random", rays), ("primary", prim)):
        h = pt.trace_closest(r, abi.PRECISION_F32)
        t0 = time.time(); h = pt.trace_closest(r, abi.PRECISION_F32); dt = time.time() - t0
        v = h["u"]
        print(tag, "steps/ray mean", v.mean(), "p50", np.median(v), "p99", np.percentile(v, 99), "max", v.max(), "hit frac", h["hit"].mean(), "tri frac", (h["prim_type"] == 2).mean(), f"call {dt*1e3:.1f} ms incl copies")
