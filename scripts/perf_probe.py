"""Time the headline workload with one or more builds of the library (A/B in one process order).
usage: perf_probe.py [--spp N] name=path.so ..."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import subprocess
spp = 128
args = sys.argv[1:]
if args and args[0] == "--spp":
    spp = int(args[1]); args = args[2:]
if len(args) == 1 and args[0].startswith("@child"):
    pass
code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
import rtx_pkg; pkg = rtx_pkg.load()
sd = pkg.scene_loader.load(os.path.join(%r, "tests/golden/scenes", sys.argv[2] + ".json"))
spp = int(sys.argv[1])
with pkg.renderer.HipPathTracer(0) as pt:
    pt.upload_scene(sd); pt.set_camera(sd.camera(1920, 1080))
    pt.set_settings(sd.settings(width=1920, height=1080, samples_per_pixel=spp, max_depth=sd.max_depth if sd.max_depth < 500 else 500))
    best = None
    for i in range(4):
        f = pt.render(); st = pt.last_stats
        best = st.seconds_main_kernel if best is None else min(best, st.seconds_main_kernel)
    rays = st.rays_closest + st.rays_shadow
    print("%%-28s %%-18s %%8.2f ms  %%8.1f Msamples/s  %%7.2f Grays/s  mean %%.5f" %% (os.environ.get("VARIANT"), sys.argv[2], best*1e3, st.samples/best/1e6, rays/best/1e9, float(np.clip(f,0,1).mean())), flush=True)
''' % (ROOT, ROOT)
scenes = os.environ.get("SCENES", "cornell_box").split(",")
for a in args:
    name, path = a.split("=")
    for sc in scenes:
        env = dict(os.environ, RTX_HIP_LIB=os.path.join(ROOT, path), VARIANT=name)
        subprocess.run([sys.executable, "-c", code, str(spp), sc], env=env, check=False, timeout=300)
