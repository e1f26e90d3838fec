"""rtx_render_device_noalloc against rtx_render_device through the C ABI: the same kernels with their chunk sums and counters in
the library's static scratch array instead of allocated buffers, so films must agree bit for bit — on two calls in a row, the
second on a smaller input (the array keeps the first call's contents), with and without statistics, and its limits must be
refused with a message."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_noalloc_render_equals_allocating_render(pkg, load_scene):
    import torch
    abi = pkg.abi
    sd = load_scene("cornell_box")
    with pkg.renderer.HipPathTracer(0) as pt:
        pt.upload_scene(sd)
        for (W, H, spp, depth, mode) in ((640, 360, 24, 12, abi.MODE_PATHTRACE), (96, 54, 5, 8, abi.MODE_PATHTRACE), (64, 36, 8, 6, abi.MODE_WHITTED)):
            pt.set_camera(sd.camera(W, H))
            pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=spp, max_depth=depth, seed=5, mode=mode))
            want = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
            got = torch.full((H, W, 3), -1.0, dtype=torch.float32, device="cuda:0")
            ref_stats = pt.render_device(want.data_ptr())
            st = pt.render_device_noalloc(got.data_ptr())
            assert (st.samples, st.rays_closest, st.rays_shadow, st.segments) == (ref_stats.samples, ref_stats.rays_closest, ref_stats.rays_shadow, ref_stats.segments)
            assert st.samples == W * H * spp
            assert np.array_equal(got.cpu().numpy().view(np.uint32), want.cpu().numpy().view(np.uint32)), (W, H, spp)
            # without statistics the call only enqueues: synchronise, then compare
            got.fill_(-1.0)
            torch.cuda.synchronize()
            assert pt.render_device_noalloc(got.data_ptr(), torch.cuda.current_stream().cuda_stream, want_stats=False) is None
            torch.cuda.synchronize()
            assert np.array_equal(got.cpu().numpy().view(np.uint32), want.cpu().numpy().view(np.uint32))
        # a shard of a larger film fits where the whole film does not
        W, H, spp = 1920, 1080, 16
        pt.set_camera(sd.camera(W, H))
        pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=spp, max_depth=8, seed=5))
        big = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        with pytest.raises(abi.RtxError, match="chunk sums"):
            pt.render_device_noalloc(big.data_ptr())
        pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=spp, max_depth=8, seed=5, shard_index=3, shard_count=16))
        rows = len(range(3, H, 16))
        part_a = torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda:0")
        part_b = torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda:0")
        pt.render_device(part_a.data_ptr())
        pt.render_device_noalloc(part_b.data_ptr())
        assert np.array_equal(part_a.cpu().numpy().view(np.uint32), part_b.cpu().numpy().view(np.uint32))
        # what needs other buffers is refused
        small = torch.zeros((36, 64, 3), dtype=torch.float32, device="cuda:0")
        pt.set_camera(sd.camera(64, 36))
        for bad, msg in ((dict(precision=abi.PRECISION_F64), "single-precision"), (dict(mode=abi.MODE_BDPT), "single-precision"), (dict(use_mnee=1), "single-precision"),
                         (dict(pipeline=abi.PIPELINE_WAVEFRONT), "fused pipeline")):
            pt.set_settings(sd.settings(width=64, height=36, samples_per_pixel=4, max_depth=6, **bad))
            with pytest.raises(abi.RtxError, match=msg):
                pt.render_device_noalloc(small.data_ptr())
        lib = pkg.abi.load_library()
        assert lib.rtx_render_device_noalloc(None, None, None, None, None) == -1


def test_noalloc_render_walks_a_tree_and_refuses_the_queue_pipeline(pkg, load_scene):
    import torch
    from conftest import soup_scene
    abi = pkg.abi
    sd = soup_scene(pkg, load_scene)   # 4002 bounded primitives: an eight-wide tree a few levels deep
    W, H = 80, 45
    with pkg.renderer.HipPathTracer(0) as pt:
        pt.upload_scene(sd)
        assert pt.bvh_info().wide_levels >= 2
        pt.set_camera(sd.camera(W, H))
        pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=6, max_depth=10, seed=11, pipeline=abi.PIPELINE_FUSED))
        a = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        b = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        pt.render_device(a.data_ptr())
        pt.render_device_noalloc(b.data_ptr())
        assert np.array_equal(a.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32))
        pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=6, max_depth=10, seed=11))   # AUTO picks the queue-based pipeline here
        with pytest.raises(abi.RtxError, match="fused pipeline"):
            pt.render_device_noalloc(b.data_ptr())


def test_noalloc_render_of_a_large_scene_waits_for_the_reference_order_tree(pkg, load_scene, monkeypatch):
    """A frame rendered ahead of the reference-order tree of a large scene needs a tie list (an allocation) and a repair pass: the
    no-allocation render takes neither, it waits for the tree, with or without statistics, and gives the eager build's film."""
    import copy
    import torch
    abi = pkg.abi
    sd = copy.copy(load_scene("cornell_box_quad"))
    sd._keep = []
    soup = pkg.synthetic.triangle_soup(300_000, seed=9, extent=2.2, size=0.08, n_materials=5)
    for k in ("v0", "v1", "v2"):
        soup[k] += (0, 2.5, -2.5)
    twin = soup[:3000].copy()                      # 3 000 triangles occur twice: exact ties in t
    twin["material_id"] = (twin["material_id"] + 2) % 5
    sd.triangle_array = np.concatenate([soup, twin])
    W, H = 160, 90
    kw = dict(width=W, height=H, samples_per_pixel=8, max_depth=6, seed=8, pipeline=abi.PIPELINE_FUSED)

    def render(env, with_stats):
        for k in ("RTX_EAGER_TREE", "RTX_TREE_DELAY_MS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with pkg.renderer.HipPathTracer(0) as pt:
            pt.upload_scene(sd)
            pt.set_camera(sd.camera(W, H))
            pt.set_settings(sd.settings(**kw))
            out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
            st = pt.render_device_noalloc(out.data_ptr(), want_stats=with_stats)
            assert st is None or st.flags == 0
            torch.cuda.synchronize()
            return out.cpu().numpy()

    want = render({"RTX_EAGER_TREE": "1"}, True)
    for with_stats in (True, False):
        got = render({"RTX_TREE_DELAY_MS": "800"}, with_stats)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_noalloc_render_recorded_into_a_graph(pkg, load_scene):
    """The stated use of the entry point: with stats = NULL it puts nothing but kernels on the stream, so a caller can record it
    into a HIP graph.  The graph is replayed several times with eager work and synchronisations of the program in between; every
    replay must give the eager film."""
    import torch
    sd = load_scene("cornell_box")
    W, H = 128, 72
    with pkg.renderer.HipPathTracer(0) as pt:
        pt.upload_scene(sd)
        pt.set_camera(sd.camera(W, H))
        pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=10, max_depth=8, seed=17))
        want = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        pt.render_device(want.data_ptr())
        got = torch.full((H, W, 3), -1.0, dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            assert pt.render_device_noalloc(got.data_ptr(), torch.cuda.current_stream().cuda_stream, want_stats=False) is None
        other = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        for i in range(3):
            got.fill_(-1.0 - i)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            assert np.array_equal(got.cpu().numpy().view(np.uint32), want.cpu().numpy().view(np.uint32)), i
            # eager work of the same context between the replays (it shares the scratch array: the replay must not depend on it)
            pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=3 + i, max_depth=4, seed=5 + i))
            pt.render_device_noalloc(other.data_ptr())
            assert float(other.sum().item()) > 0.0
