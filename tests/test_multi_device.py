"""The multi-device context of the C ABI (rtx_create_multi).

Without a GPU: the arithmetic that deals a shard's rows to the devices.  With one GPU: the whole path on a context that lists that
GPU twice or three times (every listing holds its own scene copy and renders its own share), whose gathered film must equal the
single-device film bit for bit, in host and in device memory, sharded and not."""
import ctypes as C

import numpy as np
import pytest

from conftest import ROOT  # noqa: F401


def test_rows_are_dealt_to_devices_in_turn(pkg):
    lib = pkg.abi.load_library()
    for height in (1, 5, 36, 1080):
        for sc in (1, 2, 3):
            for si in range(sc):
                parent = list(range(si, height, sc))            # film rows of the caller's shard, packed bottom-up
                for n in (1, 2, 3, 8):
                    seen = {}
                    for k in range(n):
                        idx, cnt = C.c_int(), C.c_int()
                        assert lib.rtx_multi_shard(si, sc, k, n, C.byref(idx), C.byref(cnt)) == 0
                        rows = list(range(idx.value, height, cnt.value))
                        assert len(rows) == lib.rtx_shard_rows(height, idx.value, cnt.value)
                        assert rows == parent[k::n]              # packed row j of the shard is packed row j // n of device j % n
                        for y in rows:
                            assert y not in seen
                            seen[y] = k
                    assert sorted(seen) == parent
    idx, cnt = C.c_int(), C.c_int()
    assert lib.rtx_multi_shard(0, 0, 1, 2, C.byref(idx), C.byref(cnt)) == 0 and (idx.value, cnt.value) == (1, 2)   # shard_count 0 means 1
    for bad in ((2, 2, 0, 1), (0, 1, 1, 1), (0, 1, -1, 2), (0, 1, 0, 0), (-1, 1, 0, 1)):
        assert lib.rtx_multi_shard(*bad, C.byref(idx), C.byref(cnt)) == -1
    assert lib.rtx_multi_shard(0, 1, 0, 1, None, None) == -1


def test_create_multi_rejects_bad_lists(pkg):
    lib = pkg.abi.load_library()
    ctx = C.c_void_p()
    assert lib.rtx_create_multi(C.byref(ctx), None, 1) == -1
    ids = (C.c_int * 1)(0)
    assert lib.rtx_create_multi(C.byref(ctx), ids, 0) == -1
    assert lib.rtx_create_multi(None, ids, 1) == -1


def _film(pkg, tracer, scene, **kw):
    st = scene.settings(**kw)
    tracer.set_settings(st)
    tracer.set_camera(scene.camera(st.width, st.height))
    return tracer.render().copy(), tracer.last_stats


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["cornell_fused", "soup_queues"])
def test_multi_device_film_equals_single_device_film(pkg, case):
    import torch
    if case == "cornell_fused":
        scene = pkg.scene_loader.load(ROOT + "/tests/golden/scenes/cornell_box.json")
        kw = dict(width=96, height=54, samples_per_pixel=8, max_depth=6, seed=11)
    else:
        scene = pkg.synthetic.stress_scene(3000)
        kw = dict(width=64, height=37, samples_per_pixel=4, max_depth=5, seed=5)
    with pkg.renderer.HipPathTracer(0) as one:
        one.upload_scene(scene)
        whole, st1 = _film(pkg, one, scene, **kw)
        shard1, _ = _film(pkg, one, scene, shard_index=1, shard_count=2, **kw)
    for ids in ([0], [0, 0], [0, 0, 0]):
        with pkg.renderer.HipPathTracer(device_ids=ids) as multi:
            multi.upload_scene(scene)
            got, stn = _film(pkg, multi, scene, **kw)
            assert np.array_equal(got, whole), ids
            assert stn.samples == st1.samples and stn.rays_closest == st1.rays_closest and stn.segments == st1.segments
            got1, _ = _film(pkg, multi, scene, shard_index=1, shard_count=2, **kw)
            assert np.array_equal(got1, shard1), ids
            # the same into device memory of the first device
            out = torch.full(whole.shape, -1.0, dtype=torch.float32, device="cuda:0")
            multi.set_settings(scene.settings(**kw))
            multi.render_device(out.data_ptr())
            assert np.array_equal(out.cpu().numpy(), whole), ids
            info = multi.bvh_info()
            assert info.ref_prims == scene.bounded_primitive_count()
            with pytest.raises(pkg.abi.RtxError, match="single-device"):
                multi.render_device_noalloc(out.data_ptr())


@pytest.mark.gpu
def test_multi_device_more_devices_than_rows(pkg):
    scene = pkg.scene_loader.load(ROOT + "/tests/golden/scenes/cornell_box.json")
    kw = dict(width=16, height=2, samples_per_pixel=2, max_depth=3, seed=3)
    with pkg.renderer.HipPathTracer(0) as one:
        one.upload_scene(scene)
        whole, _ = _film(pkg, one, scene, **kw)
    with pkg.renderer.HipPathTracer(device_ids=[0, 0, 0]) as multi:
        multi.upload_scene(scene)
        got, _ = _film(pkg, multi, scene, **kw)
    assert np.array_equal(got, whole)
