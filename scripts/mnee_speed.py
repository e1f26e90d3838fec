"""Speed of use_mnee = 1 on the headline scene (1920x1080, 16 spp single precision / 4 spp double precision, default settings)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_pkg
pkg = rtx_pkg.load(); abi = pkg.abi
sd = pkg.scene_loader.load(os.path.join(ROOT, "tests", "golden", "scenes", "cornell_box.json"))
with pkg.renderer.HipPathTracer(0) as pt:
    pt.upload_scene(sd)
    W, H = 1920, 1080
    pt.set_camera(sd.camera(W, H))
    for tag, prec, spp in (("f32", abi.PRECISION_F32, 16), ("f64", abi.PRECISION_F64, 4)):
        pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=spp, use_nee=1, use_mis=1, use_mnee=1, precision=prec))
        best = None
        for _ in range(3):
            pt.render()
            s = pt.last_stats
            best = s.seconds_render if best is None or s.seconds_render < best else best
        print(json.dumps({"mnee": tag, "spp": spp, "ms": best * 1e3, "msamples_per_s": W * H * spp / best / 1e6, "closest_per_sample": s.rays_closest / s.samples}))
