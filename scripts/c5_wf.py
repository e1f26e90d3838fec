import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_pkg
pkg = rtx_pkg.load(); abi = pkg.abi
n = int(sys.argv[1]); spp = int(sys.argv[2]); pipe = abi.PIPELINE_WAVEFRONT if (len(sys.argv) < 4 or sys.argv[3] == "wf") else abi.PIPELINE_FUSED
sd = pkg.synthetic.stress_scene(n)
with pkg.renderer.HipPathTracer(0) as pt:
    pt.upload_scene(sd); pt.set_camera(sd.camera(1920, 1080))
    pt.set_settings(sd.settings(width=1920, height=1080, samples_per_pixel=spp, max_depth=8, pipeline=pipe))
    for _ in range(2):
        pt.render(); st = pt.last_stats
        print(f"{st.seconds_main_kernel*1e3:.1f} ms  {(st.rays_closest+st.rays_shadow)/st.seconds_main_kernel/1e9:.3f} Grays/s closest {st.rays_closest} shadow {st.rays_shadow}", flush=True)
