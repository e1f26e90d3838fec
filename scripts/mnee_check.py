"""Single-precision MNEE against the double-precision parity mode on the same random numbers, and its speed."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rtx_pkg
pkg = rtx_pkg.load(); abi = pkg.abi
SCENES = os.path.join(ROOT, "tests", "golden", "scenes")
for name in sys.argv[1:] or ["cornell_box", "cornell_box_quad", "caustics_test"]:
    sd = pkg.scene_loader.load(os.path.join(SCENES, name + ".json"))
    with pkg.renderer.HipPathTracer(0) as pt:
        pt.upload_scene(sd)
        W, H, spp = 160, 96, 64
        pt.set_camera(sd.camera(W, H))
        over = dict(width=W, height=H, samples_per_pixel=spp, max_depth=min(sd.max_depth, 12), seed=47, use_nee=1, use_mis=1, rng=abi.RNG_PHILOX)
        films = {}
        for tag, kw in (("f64_mnee", dict(precision=abi.PRECISION_F64, use_mnee=1)), ("f32_mnee", dict(precision=abi.PRECISION_F32, use_mnee=1)),
                        ("f64_off", dict(precision=abi.PRECISION_F64, use_mnee=0)), ("f32_off", dict(precision=abi.PRECISION_F32, use_mnee=0))):
            pt.set_settings(sd.settings(**over, **kw))
            films[tag] = pt.render().astype(np.float64).copy()
        def bad(a, b): return float((np.abs(a - b) > 2e-3 * (1.0 + np.abs(b))).any(axis=2).mean())
        out = {"scene": name, "bad_f32_vs_f64_mnee": bad(films["f32_mnee"], films["f64_mnee"]), "bad_f32_vs_f64_off": bad(films["f32_off"], films["f64_off"]),
               "mean_f64_mnee": films["f64_mnee"].mean(), "mean_f32_mnee": films["f32_mnee"].mean(), "mean_f64_off": films["f64_off"].mean(),
               "mnee_part_f64": (films["f64_mnee"] - films["f64_off"]).mean(), "mnee_part_f32": (films["f32_mnee"] - films["f32_off"]).mean(),
               "pixels_with_mnee_f64": float(((films["f64_mnee"] - films["f64_off"]).max(axis=2) > 1e-4).mean()),
               "pixels_with_mnee_f32": float(((films["f32_mnee"] - films["f32_off"]).max(axis=2) > 1e-4).mean()),
               "pixels_with_mnee_both": float((((films["f32_mnee"] - films["f32_off"]).max(axis=2) > 1e-4) & ((films["f64_mnee"] - films["f64_off"]).max(axis=2) > 1e-4)).mean())}
        W, H = 1920, 1080
        pt.set_camera(sd.camera(W, H))
        for tag, kw, spp in (("f32_mnee", dict(precision=abi.PRECISION_F32, use_mnee=1), 8), ("f64_mnee", dict(precision=abi.PRECISION_F64, use_mnee=1), 1)):
            pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=spp, max_depth=sd.max_depth, seed=1234, use_nee=1, use_mis=1, **kw))
            pt.render(); pt.render()
            st = pt.last_stats
            out["msamples_per_s_" + tag] = round(st.samples / st.seconds_render / 1e6, 2)
            out["closest_per_sample_" + tag] = round(st.rays_closest / st.samples, 2)
        print(json.dumps(out), flush=True)
