"""Runs the measurement configurations C1-C5 once each on cuda:0 and prints one JSON line per run.

  python scripts/run_configs.py [c2 c3 c4 c5 ...] [--repeat N] [--spp N] [--pipeline auto|fused|wavefront] [--mnee] [--f64]
C4 at its full 3840x2160x4096 size takes several seconds per scene; C5 builds a 10 M triangle scene."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rtx_pkg
pkg = rtx_pkg.load(); abi = pkg.abi
SCENES = os.path.join(ROOT, "tests", "golden", "scenes")

CONFIGS = {
    "c1": [("demo", dict(width=2560, height=1440, samples_per_pixel=64, max_depth=50, mode=0))],   # the scene file's own mode: Whitted
    "c2": [("cornell_box", dict(width=1920, height=1080, samples_per_pixel=1024, max_depth=500))],
    "c3": [("cornell_box_quad", dict(width=800, height=800, samples_per_pixel=128, max_depth=50)),
           ("caustics_test", dict(width=800, height=800, samples_per_pixel=128, max_depth=16))],
    "c3bdpt": [("caustics_test", dict(width=800, height=800, samples_per_pixel=128, max_depth=16, mode=2))],   # the scene file's own mode: BDPT
    "c2photon": [("cornell_box", dict(width=1920, height=1080, samples_per_pixel=16, max_depth=500, mode=6))],   # PathPhoton with the reference's defaults (50000 photons, 100 gathered, radius 0.1)
    "c4env": [("environment_demo", dict(width=3840, height=2160, samples_per_pixel=256))],
    "c4": [("environment_demo", dict(width=3840, height=2160, samples_per_pixel=4096)),
           ("earth_demo", dict(width=3840, height=2160, samples_per_pixel=4096))],
    "c5": [("stress10m", dict(width=1920, height=1080, samples_per_pixel=64, max_depth=8))],
}

def main():
    args = [a for a in sys.argv[1:] if a in CONFIGS]
    spp = int(sys.argv[sys.argv.index("--spp") + 1]) if "--spp" in sys.argv else None
    repeat = int(sys.argv[sys.argv.index("--repeat") + 1]) if "--repeat" in sys.argv else 2
    pipes = {"auto": abi.PIPELINE_AUTO, "fused": abi.PIPELINE_FUSED, "wavefront": abi.PIPELINE_WAVEFRONT}
    pipe = pipes[sys.argv[sys.argv.index("--pipeline") + 1]] if "--pipeline" in sys.argv else abi.PIPELINE_AUTO
    for cfg in (args or ["c1", "c2", "c3", "c4", "c5"]):
        for name, over in CONFIGS[cfg]:
            if spp: over = dict(over, samples_per_pixel=spp)
            t0 = time.time()
            sd = pkg.synthetic.stress_scene(10_000_000) if name == "stress10m" else pkg.scene_loader.load(os.path.join(SCENES, name + ".json"))
            t_load = time.time() - t0
            with pkg.renderer.HipPathTracer(0) as pt:
                t0 = time.time(); pt.upload_scene(sd); t_upload = time.time() - t0
                info = pt.bvh_info() if hasattr(pt, "bvh_info") else None
                pt.set_camera(sd.camera(over["width"], over["height"]))
                pt.set_settings(sd.settings(use_nee=1, use_mis=1, use_mnee=int('--mnee' in sys.argv), pipeline=pipe, precision=abi.PRECISION_F64 if '--f64' in sys.argv else abi.PRECISION_F32, **over))
                best = None
                for _ in range(repeat):
                    film = pt.render(); st = pt.last_stats
                    if best is None or st.seconds_render < best.seconds_render: best = st
                st = best
                rays = st.rays_closest + st.rays_shadow
                print(json.dumps({
                    "config": cfg, "scene": name, **over, "pipeline_used": {1: "fused", 2: "wavefront"}.get(st.pipeline_used, st.pipeline_used), "precision_used": {0: "f32", 1: "f64"}.get(st.precision_used), "mnee": int("--mnee" in sys.argv),
                    "ms_render": round(st.seconds_render * 1e3, 3), "ms_main_kernels": round(st.seconds_main_kernel * 1e3, 3),
                    "msamples_per_s": round(st.samples / st.seconds_render / 1e6, 2), "mrays_per_s": round(rays / st.seconds_render / 1e6, 2),
                    "segments_per_sample": round(st.segments / st.samples, 3), "shadow_per_sample": round(st.rays_shadow / st.samples, 3),
                    "kernel_launches": st.kernel_launches, "film_mean_clamped": float(np.clip(film, 0, 1).mean()),
                    "scene_load_s": round(t_load, 3), "upload_and_bvh_s": round(t_upload, 3)}), flush=True)

if __name__ == "__main__":
    main()
