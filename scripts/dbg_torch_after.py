import os, sys, ctypes, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rtx_pkg; pkg = rtx_pkg.load(); abi = pkg.abi
use_oracle = int(sys.argv[1]); keep = int(sys.argv[2])
hip = ctypes.CDLL("libamdhip64.so")
def count():
    n = ctypes.c_int(0); rc = hip.hipGetDeviceCount(ctypes.byref(n)); return rc, n.value
alive = []
for name in ("cornell_box", "cornell_box", "cornell_box_quad", "caustics_test", "kitchen_sink", "disk_room"):
    sd = pkg.scene_loader.load(os.path.join(ROOT, "tests/golden/scenes", name + ".json"))
    st = sd.settings(width=40, height=24, samples_per_pixel=6, max_depth=12, use_mnee=1, use_mis=0)
    if use_oracle:
        import oracle_py
        oracle_py.OracleScene(sd).render(sd.camera(40, 24), st, oracle_py.RNG_COUNTER, threads=16)
    pt = pkg.renderer.HipPathTracer(0)
    pt.upload_scene(sd); pt.set_camera(sd.camera(40, 24)); pt.set_settings(st)
    f = pt.render(); print(name, "render ok", f.mean(), count(), flush=True)
    if keep: alive.append(pt)
    else: pt.close() if hasattr(pt, "close") else None
import torch
print("torch device count", torch.cuda.device_count(), count(), flush=True)
try:
    torch.cuda.init(); print("torch ok", torch.zeros(1, device="cuda"))
except Exception as e:
    print("torch failed:", e, count())
