import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_pkg; pkg = rtx_pkg.load(); abi = pkg.abi
name = sys.argv[1] if len(sys.argv) > 1 else "kitchen_sink"
sd = pkg.scene_loader.load(os.path.join(ROOT, "tests/golden/scenes", name + ".json"))
W, H = 80, 45
with pkg.renderer.HipPathTracer(0) as pt:
    pt.upload_scene(sd); pt.set_camera(sd.camera(W, H))
    res = {}
    for tag, kw in (("f64 fused", dict(precision=abi.PRECISION_F64, pipeline=abi.PIPELINE_FUSED)), ("f32 fused", dict(pipeline=abi.PIPELINE_FUSED)), ("f32 wave", dict(pipeline=abi.PIPELINE_WAVEFRONT))):
        for depth in (1, 2, 30):
            pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=12, max_depth=depth, seed=77, **kw))
            f = pt.render().astype(np.float64); st = pt.last_stats
            res[(tag, depth)] = f
            print(tag, "depth", depth, "mean", f.mean(axis=(0,1)), "closest", st.rays_closest, "shadow", st.rays_shadow, "seg", st.segments)
    for depth in (1, 2, 30):
        a, b, c = res[("f64 fused", depth)], res[("f32 fused", depth)], res[("f32 wave", depth)]
        print("depth", depth, "f32fused-f64 max", np.abs(b - a).max(), "wave-f32fused max", np.abs(c - b).max(), "frac", (np.abs(c-b) > 1e-3*(1+np.abs(b))).mean())
