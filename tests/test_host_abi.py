"""Host logic that needs no GPU: the shared library loads and exports exactly what
include/rtx_hip.h declares, struct layouts agree between C and ctypes, error reporting works
without a device, and the Python mirrors of the post-process follow the reference formulas."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "rtx_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtx_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.abi.load_library()
    names = declared_functions()
    assert sorted(pkg.abi.EXPORTS) == names
    for n in names:
        assert getattr(lib, n) is not None


def test_struct_layouts_match_the_c_header(pkg):
    abi = pkg.abi
    pairs = {"rtx_sphere": abi.Sphere, "rtx_plane": abi.Plane, "rtx_triangle": abi.Triangle, "rtx_box": abi.Box,
             "rtx_point_light": abi.PointLight, "rtx_quad_light": abi.QuadLight, "rtx_disk_light": abi.DiskLight,
             "rtx_material": abi.Material, "rtx_image": abi.Image, "rtx_environment": abi.Environment,
             "rtx_scene_desc": abi.SceneDesc, "rtx_camera": abi.Camera, "rtx_settings": abi.Settings, "rtx_stats": abi.Stats,
             "rtx_hit": abi.Hit, "rtx_bvh_info": abi.BvhInfo}
    src = '#include <stdio.h>\n#include "rtx_hip.h"\nint main(void){\n' + "".join(
        f'printf("{n} %zu\\n", sizeof({n}));\n' for n in pairs) + \
        'printf("off_seed %zu\\n", __builtin_offsetof(rtx_settings, seed));\n' \
        'printf("off_t %zu\\n", __builtin_offsetof(rtx_hit, t));\n' \
        'printf("off_env %zu\\n", __builtin_offsetof(rtx_scene_desc, environment));\nreturn 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = dict(l.split() for l in subprocess.check_output([exe], text=True).splitlines())
    for n, t in pairs.items():
        assert int(out[n]) == C.sizeof(t), n
    assert int(out["off_seed"]) == abi.Settings.seed.offset
    assert int(out["off_t"]) == abi.Hit.t.offset
    assert int(out["off_env"]) == abi.SceneDesc.environment.offset
    assert pkg.scene_loader.TRIANGLE_DTYPE.itemsize == C.sizeof(abi.Triangle)


def test_shard_rows(pkg):
    lib = pkg.abi.load_library()
    for h in (1, 7, 36, 1080):
        for n in (1, 2, 3, 8):
            rows = [lib.rtx_shard_rows(h, k, n) for k in range(n)]
            assert sum(rows) == h and rows == [len(range(k, h, n)) for k in range(n)]
    assert lib.rtx_shard_rows(10, 3, 3) == 0 and lib.rtx_shard_rows(10, -1, 3) == 0 and lib.rtx_shard_rows(10, 0, 0) == 0


def test_errors_are_codes_and_messages_not_crashes(pkg):
    """Without a visible device every entry point must fail with a code and a message (the
    reference's GPU plug-in only printf()s, engine/cuda/cuda_common.cuh:124-131)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible; the no-device behaviour is covered on CPU-only hosts")
    lib = pkg.abi.load_library()
    ctx = C.c_void_p()
    rc = lib.rtx_create(C.byref(ctx), 0)
    assert rc == -2 and b"no HIP device" in lib.rtx_last_error()
    assert lib.rtx_selftest(0) == -2
    assert lib.rtx_render(None, None, None, None) == -1
    assert lib.rtx_upload_scene(None, None) == -1
    with pytest.raises(pkg.abi.RtxError):
        pkg.renderer.HipPathTracer(0)
    assert pkg.renderer.test_hip(0) is False


def test_missing_library_fails_loudly(pkg):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.abi.load_library("/nonexistent/librtx_hip.so")


def test_tone_map_and_quantize_follow_the_reference_formulas(pkg):
    """engine/renderer.cpp:102-198 (operators), :232-258 (sqrt gamma, 255.999, flip)."""
    r = pkg.renderer
    sl = pkg.scene_loader
    x = np.array([[[0.0, 0.18, 1.0]], [[2.5, 10.0, 100.0]]])
    assert np.allclose(r.tone_map(x, sl.TM_NONE), np.clip(x, 0, 1))
    assert np.allclose(r.tone_map(x, sl.TM_REINHARD), x / (1 + x))
    a = r.tone_map(x, sl.TM_ACES, exposure=2.0)
    e = 2.0 * x
    assert np.allclose(a, np.clip((e * (2.51 * e + 0.03)) / (e * (2.43 * e + 0.59) + 0.14), 0, 1))
    u = r.tone_map(np.array([[[11.2 / 2.0] * 3]]), sl.TM_UNCHARTED2)
    assert np.allclose(u, 1.0)
    q = r.quantize(np.array([[[0.0, 0.25, 1.0]], [[1.0, 1.0, 1.0]]]))
    assert q.dtype == np.uint8 and q.shape == (2, 1, 3)
    assert q[1, 0].tolist() == [0, 127, 255] and q[0, 0].tolist() == [255, 255, 255]   # row 0 of the output is the TOP row


def test_triangle_soup_generator_is_stable(pkg):
    a = pkg.synthetic.triangle_soup(1000, seed=42)
    b = pkg.synthetic.triangle_soup(1000, seed=42)
    assert np.array_equal(a["v0"], b["v0"]) and a["material_id"].max() == 15
    assert np.abs(a["v0"]).max() <= 10.05 and np.abs(a["v1"] - a["v0"]).max() <= 0.1
    # pinned values: the fixtures under tests/golden/hits/soup.npz depend on this exact stream
    s = pkg.synthetic.triangle_soup(3, seed=7, extent=2.2, size=0.25, n_materials=5)
    assert s["v0"].shape == (3, 3)
    sd = pkg.synthetic.stress_scene(100)
    assert sd.bounded_primitive_count() == 100 and len(sd.planes) == 5 and len(sd.quad_lights) == 1 and len(sd.materials) == 16
