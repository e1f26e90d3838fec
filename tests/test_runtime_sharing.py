"""One HIP runtime per process, whichever of the library and torch is loaded first.

torch's ROCm wheels bundle their own HIP runtime and link it by file name; the dynamic loader does not match that name with a
runtime loaded earlier under its SONAME from /opt/rocm, so a process that loaded librtx_hip.so before `import torch` held two
runtimes, and torch's could not initialise the device ("No HIP GPUs are available").  abi.load_library now loads torch's copy
first when torch is installed; these tests run each import order in a fresh interpreter."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

COUNT_RUNTIMES = """
import sys
sys.path.insert(0, %r)
order = sys.argv[1]
if order == "torch_first":
    import torch
import rtx_pkg
pkg = rtx_pkg.load()
pkg.abi.load_library()
if order == "library_first":
    import torch
maps = open("/proc/self/maps").read().splitlines()
print("RUNTIMES", len({l.split()[-1] for l in maps if "libamdhip64" in l}), len({l.split()[-1] for l in maps if "libhsa-runtime64" in l}))
""" % ROOT


@pytest.mark.parametrize("order", ["library_first", "torch_first"])
def test_one_hip_runtime_is_mapped(order):
    out = subprocess.run([sys.executable, "-c", COUNT_RUNTIMES, order], capture_output=True, text=True, check=True).stdout
    assert "RUNTIMES 1 1" in out, out


LIBRARY_THEN_TORCH = """
import os, sys
sys.path.insert(0, %r)
import numpy as np
import rtx_pkg
pkg = rtx_pkg.load()
sd = pkg.scene_loader.load(os.path.join(%r, "tests", "golden", "scenes", "cornell_box.json"))
for _ in range(3):
    film, st = pkg.renderer.render_scene(sd, width=64, height=36, samples_per_pixel=4, max_depth=8, seed=1)
assert st.samples == 64 * 36 * 4
import torch
assert torch.cuda.is_available()
x = torch.arange(8, device="cuda", dtype=torch.float32)
dev = torch.zeros((36, 64, 3), device="cuda")
with pkg.renderer.HipPathTracer(0) as pt:
    pt.set_settings(sd.settings(width=64, height=36, samples_per_pixel=4, max_depth=8, seed=1))
    pt.upload_scene(sd)
    pt.set_camera(sd.camera(64, 36))
    pt.render_device(dev.data_ptr())
assert np.array_equal(dev.cpu().numpy(), film) and float(x.sum()) == 28.0
print("SHARED_RUNTIME_OK")
""" % (ROOT, ROOT)


@pytest.mark.gpu
def test_torch_initialises_after_the_library_has_rendered():
    p = subprocess.run([sys.executable, "-c", LIBRARY_THEN_TORCH], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "SHARED_RUNTIME_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-2000:])
