"""Additions of the C ABI that change HOW a render runs, never its result: the two-halves render (rtx_render_device_async /
rtx_render_wait), the smaller pools the queue pipeline falls back to, and the RCCL film gather of a multi-device context.  Every film must
equal the plain one bit for bit."""
import os

import numpy as np
import pytest

from conftest import soup_scene

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.cpu().numpy().view(np.uint32)


def test_async_render_equals_synchronous_render(pkg, load_scene):
    import torch
    abi = pkg.abi
    stream = torch.cuda.current_stream().cuda_stream
    cases = [(load_scene("cornell_box"), dict(width=160, height=90, samples_per_pixel=12, max_depth=10, seed=3)),
             (soup_scene(pkg, load_scene), dict(width=80, height=45, samples_per_pixel=6, max_depth=8, seed=4))]   # fused, then the queue pipeline
    tracers = []
    for sd, kw in cases:
        pt = pkg.renderer.HipPathTracer(0)
        pt.upload_scene(sd)
        pt.set_camera(sd.camera(kw["width"], kw["height"]))
        pt.set_settings(sd.settings(**kw))
        tracers.append(pt)
    want, ref_stats = [], []
    for pt, (sd, kw) in zip(tracers, cases):
        t = torch.zeros((kw["height"], kw["width"], 3), dtype=torch.float32, device="cuda:0")
        ref_stats.append(pt.render_device(t.data_ptr(), stream))
        want.append(t)
    for _ in range(2):
        got = [torch.full_like(w, -1.0) for w in want]
        for pt, g in zip(tracers, got):          # both renders are enqueued before either is waited for
            pt.render_device_async(g.data_ptr(), stream)
        stats = [pt.render_wait() for pt in tracers]
        for g, w, s, r in zip(got, want, stats, ref_stats):
            assert np.array_equal(_bits(g), _bits(w))
            assert (s.samples, s.rays_closest, s.rays_shadow, s.segments) == (r.samples, r.rays_closest, r.rays_shadow, r.segments)
            assert s.seconds_render > 0 and s.seconds_total >= s.seconds_render * 0.5
    # a render that was never collected is waited for by the next call on the context; waiting twice is an error
    pt = tracers[0]
    g = torch.full_like(want[0], -1.0)
    pt.render_device_async(g.data_ptr(), stream)
    again = torch.full_like(want[0], -1.0)
    pt.render_device(again.data_ptr(), stream)
    assert np.array_equal(_bits(g), _bits(want[0])) and np.array_equal(_bits(again), _bits(want[0]))
    with pytest.raises(abi.RtxError, match="no render in flight"):
        pt.render_wait()
    lib = abi.load_library()
    assert lib.rtx_render_device_async(pt._ctx, None, None, None) == -1
    # a stream of the caller's own
    side = torch.cuda.Stream()
    g.fill_(-1.0)
    torch.cuda.synchronize()
    pt.render_device_async(g.data_ptr(), side.cuda_stream)
    pt.render_wait()
    assert np.array_equal(_bits(g), _bits(want[0]))
    for pt in tracers:
        pt.close()


def _tree_flags(pkg, st):
    """rtx_stats.flags without the bit that only says how the queue pipeline scheduled its two walks."""
    return st.flags & ~pkg.abi.STAT_WALKS_SIDE_BY_SIDE


def _queue_film(pkg, sd, env, **kw):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with pkg.renderer.HipPathTracer(0) as pt:
            pt.upload_scene(sd)                      # (RTX_POOL_MULT is read at the render)
            pt.set_camera(sd.camera(kw["width"], kw["height"]))
            pt.set_settings(sd.settings(pipeline=pkg.abi.PIPELINE_WAVEFRONT, **kw))
            film = pt.render().copy()
            return film, pt.last_stats
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_a_smaller_pool_and_material_queues_give_the_same_film(pkg, load_scene):
    """The queue pipeline halves its pool of path slots when the device cannot give the full one (RTX_POOL_MULT caps the first try):
    slots are reused more often, the film is the same."""
    sd = soup_scene(pkg, load_scene, n=60000)
    kw = dict(width=640, height=360, samples_per_pixel=16, max_depth=6, seed=2)
    film, st = _queue_film(pkg, sd, {"RTX_POOL_MULT": "1"}, **kw)
    want, st0 = _queue_film(pkg, sd, {}, **kw)
    assert np.array_equal(film.view(np.uint32), want.view(np.uint32))
    assert (st.samples, st.rays_closest, st.rays_shadow, st.segments) == (st0.samples, st0.rays_closest, st0.rays_shadow, st0.segments)
    # material queues (an experiment kept for its measurement): the shade kernel takes its entries ordered by material class
    film, st = _queue_film(pkg, sd, {"RTX_MATERIAL_SORT": "1"}, **kw)
    assert np.array_equal(film.view(np.uint32), want.view(np.uint32))
    assert (st.samples, st.rays_closest, st.rays_shadow, st.segments) == (st0.samples, st0.rays_closest, st0.rays_shadow, st0.segments)


def test_the_two_walks_of_a_turn_side_by_side_or_one_after_the_other(pkg, load_scene):
    """A small frame runs the any-hit walk of every turn on a stream of its own, beside the closest-hit walk (the statistics say
    so); RTX_WALK_OVERLAP=0 puts them one after the other again.  Same film, same counters."""
    abi = pkg.abi
    sd = soup_scene(pkg, load_scene, n=60000)
    kw = dict(width=640, height=360, samples_per_pixel=16, max_depth=6, seed=2)
    beside, st1 = _queue_film(pkg, sd, {}, **kw)
    serial, st0 = _queue_film(pkg, sd, {"RTX_WALK_OVERLAP": "0"}, **kw)
    assert st1.flags & abi.STAT_WALKS_SIDE_BY_SIDE and not st0.flags & abi.STAT_WALKS_SIDE_BY_SIDE
    assert np.array_equal(beside.view(np.uint32), serial.view(np.uint32))
    assert (st1.samples, st1.rays_closest, st1.rays_shadow, st1.segments) == (st0.samples, st0.rays_closest, st0.rays_shadow, st0.segments)
    assert st1.rays_shadow > 0 and st1.seconds_extend > 0


def test_rccl_gather_of_a_multi_device_context(pkg, load_scene):
    import torch
    abi = pkg.abi
    sd = load_scene("cornell_box")
    kw = dict(width=96, height=54, samples_per_pixel=8, max_depth=6, seed=11)
    with pkg.renderer.HipPathTracer(0) as one:
        one.upload_scene(sd)
        one.set_camera(sd.camera(96, 54))
        one.set_settings(sd.settings(**kw))
        whole = one.render().copy()
        with pytest.raises(abi.RtxError, match="rtx_create_multi"):
            one.set_gather(abi.GATHER_RCCL)
    with pkg.renderer.HipPathTracer(device_ids=[0]) as multi:   # one device: the communicator is made, nothing needs to be sent
        multi.set_gather(abi.GATHER_RCCL)
        multi.upload_scene(sd)
        multi.set_camera(sd.camera(96, 54))
        multi.set_settings(sd.settings(**kw))
        out = torch.full(whole.shape, -1.0, dtype=torch.float32, device="cuda:0")
        multi.render_device(out.data_ptr())
        assert np.array_equal(out.cpu().numpy(), whole)
        multi.set_gather(abi.GATHER_PEER_COPY)
        with pytest.raises(abi.RtxError, match="gather mode"):
            multi.set_gather(7)
    with pkg.renderer.HipPathTracer(device_ids=[0, 0]) as twice:
        with pytest.raises(abi.RtxError, match="listed twice"):
            twice.set_gather(abi.GATHER_RCCL)
        twice.upload_scene(sd)                                    # the context still works with peer copies
        twice.set_camera(sd.camera(96, 54))
        twice.set_settings(sd.settings(**kw))
        assert np.array_equal(twice.render(), whole)


def _big_soup(pkg, load_scene, n, duplicates=0):
    """cornell_box_quad + n small triangles (enough for rtx_upload_scene to build the reference-order tree in the background);
    `duplicates` of them occur twice, the copy with another material: every hit on such a pair is an exact tie in t."""
    import copy
    sd = copy.copy(load_scene("cornell_box_quad"))
    sd._keep = []
    soup = pkg.synthetic.triangle_soup(n, seed=9, extent=2.2, size=0.08, n_materials=5)
    for k in ("v0", "v1", "v2"):
        soup[k] += (0, 2.5, -2.5)
    if duplicates:
        twin = soup[:duplicates].copy()
        twin["material_id"] = (twin["material_id"] + 2) % 5
        soup = np.concatenate([soup, twin])
    sd.triangle_array = soup
    return sd


@pytest.mark.parametrize("pipeline", ["queues", "fused"])
def test_render_before_the_reference_order_tree_arrives(pipeline, pkg, load_scene):
    abi = pkg.abi
    kw = dict(width=320, height=180, samples_per_pixel=8, max_depth=6, seed=8,
              pipeline=abi.PIPELINE_WAVEFRONT if pipeline == "queues" else abi.PIPELINE_FUSED)

    def render(sd, env, after=None):
        old = {k: os.environ.get(k) for k in ("RTX_EAGER_TREE", "RTX_TREE_DELAY_MS")}
        for k in old:
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            with pkg.renderer.HipPathTracer(0) as pt:
                pt.upload_scene(sd)
                pt.set_camera(sd.camera(kw["width"], kw["height"]))
                pt.set_settings(sd.settings(**kw))
                film = pt.render().copy()
                flags = _tree_flags(pkg, pt.last_stats)
                more = after(pt) if after else None
                return film, flags, more
        finally:
            for k, v in old.items():
                os.environ.pop(k, None)
                if v is not None:
                    os.environ[k] = v

    plain = _big_soup(pkg, load_scene, 300_000)
    want, flags, _ = render(plain, {"RTX_EAGER_TREE": "1"})
    assert flags == 0

    def second(pt):
        info = pt.bvh_info()                 # waits for the tree
        assert info.ref_prims == 300_002 and info.build_seconds > 0   # the soup and the two spheres of the box
        film = pt.render().copy()
        return film, _tree_flags(pkg, pt.last_stats)
    got, flags, (again, flags2) = render(plain, {"RTX_TREE_DELAY_MS": "1500"}, second)
    assert flags == abi.STAT_BEFORE_TREE                                 # rendered without the tree, and no tie occurred
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert flags2 == 0 and np.array_equal(again.view(np.uint32), want.view(np.uint32))

    # a few triangles occur twice: the work items in which a ray ties between the two are rendered again when the order is known
    twins = _big_soup(pkg, load_scene, 300_000, duplicates=3_000)
    want2, flags, _ = render(twins, {"RTX_EAGER_TREE": "1"})
    assert flags == 0 and not np.array_equal(want2, want)
    got2, flags, _ = render(twins, {"RTX_TREE_DELAY_MS": "1500"})
    assert flags == abi.STAT_BEFORE_TREE | abi.STAT_TIES_REPAIRED
    assert np.array_equal(got2.view(np.uint32), want2.view(np.uint32))
    # every triangle occurs twice: more ties than the list holds, the whole frame is rendered again
    twins = _big_soup(pkg, load_scene, 300_000, duplicates=300_000)
    want3, flags, _ = render(twins, {"RTX_EAGER_TREE": "1"})
    assert flags == 0
    got3, flags, _ = render(twins, {"RTX_TREE_DELAY_MS": "1500"})
    assert flags == abi.STAT_BEFORE_TREE | abi.STAT_RENDERED_TWICE
    assert np.array_equal(got3.view(np.uint32), want3.view(np.uint32))


def test_async_render_of_a_large_scene_is_complete_in_stream_order(pkg, load_scene, monkeypatch):
    """rtx_render_device_async promises the film complete where the stream stands: work the caller queues behind it (here a copy)
    must see the final film even when the call comes before the reference-order tree of a large scene has arrived - the call
    waits for the tree instead of rendering ahead of it (a frame rendered ahead has work items rendered again at collection)."""
    import torch
    abi = pkg.abi
    kw = dict(width=320, height=180, samples_per_pixel=8, max_depth=6, seed=8, pipeline=abi.PIPELINE_WAVEFRONT)
    twins = _big_soup(pkg, load_scene, 300_000, duplicates=3_000)
    stream = torch.cuda.current_stream().cuda_stream

    def render(env):
        for k in ("RTX_EAGER_TREE", "RTX_TREE_DELAY_MS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with pkg.renderer.HipPathTracer(0) as pt:
            pt.upload_scene(twins)
            pt.set_camera(twins.camera(kw["width"], kw["height"]))
            pt.set_settings(twins.settings(**kw))
            out = torch.full((kw["height"], kw["width"], 3), -1.0, dtype=torch.float32, device="cuda:0")
            pt.render_device_async(out.data_ptr(), stream)
            behind = out.clone()              # queued behind the render, before it is collected
            st = pt.render_wait()
            return behind.cpu().numpy(), out.cpu().numpy(), _tree_flags(pkg, st)

    want, want_out, flags = render({"RTX_EAGER_TREE": "1"})
    assert flags == 0 and np.array_equal(want.view(np.uint32), want_out.view(np.uint32))
    behind, out, flags = render({"RTX_TREE_DELAY_MS": "800"})
    assert flags == 0                                                   # (not rendered ahead of the tree)
    assert np.array_equal(behind.view(np.uint32), want.view(np.uint32)) and np.array_equal(out.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("pipeline", ["queues", "fused"])
def test_multi_device_context_renders_before_the_reference_order_tree_arrives(pipeline, pkg, load_scene, monkeypatch):
    """rtx_create_multi with a large scene: the one background job finishes the reference-order tree and copies it to every device
    of the context; frames rendered meanwhile repair their tied work items per device.  (One GPU listed twice: two contexts, two
    copies, the peer-copy gather.)"""
    abi = pkg.abi
    kw = dict(width=320, height=180, samples_per_pixel=8, max_depth=6, seed=8, pipeline=abi.PIPELINE_WAVEFRONT if pipeline == "queues" else abi.PIPELINE_FUSED)
    twins = _big_soup(pkg, load_scene, 300_000, duplicates=3_000)

    def render(device_ids, env):
        for k in ("RTX_EAGER_TREE", "RTX_TREE_DELAY_MS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with (pkg.renderer.HipPathTracer(device_ids=device_ids) if device_ids else pkg.renderer.HipPathTracer(0)) as pt:
            pt.upload_scene(twins)
            pt.set_camera(twins.camera(kw["width"], kw["height"]))
            pt.set_settings(twins.settings(**kw))
            first = pt.render().copy()
            flags = _tree_flags(pkg, pt.last_stats)
            second = pt.render().copy()
            return first, flags, second, _tree_flags(pkg, pt.last_stats)

    want, flags, _, _ = render(None, {"RTX_EAGER_TREE": "1"})
    assert flags == 0
    first, flags, second, flags2 = render([0, 0], {"RTX_TREE_DELAY_MS": "1500"})
    assert flags & abi.STAT_BEFORE_TREE and flags & abi.STAT_TIES_REPAIRED
    assert np.array_equal(first.view(np.uint32), want.view(np.uint32))
    assert flags2 == 0 and np.array_equal(second.view(np.uint32), want.view(np.uint32))


def test_a_scene_replaced_or_dropped_while_its_reference_order_tree_is_being_built(pkg, load_scene, monkeypatch):
    """rtx_upload_scene again, or rtx_destroy, while the background job of the previous upload is still at work: the job is waited
    for (it writes into buffers of the scene it belongs to), nothing of it reaches the new scene."""
    abi = pkg.abi
    big = _big_soup(pkg, load_scene, 300_000)
    small = load_scene("cornell_box")
    kw = dict(width=96, height=54, samples_per_pixel=8, max_depth=6, seed=3)
    with pkg.renderer.HipPathTracer(0) as pt:
        pt.upload_scene(small)
        pt.set_camera(small.camera(96, 54))
        pt.set_settings(small.settings(**kw))
        want = pt.render().copy()
    monkeypatch.setenv("RTX_TREE_DELAY_MS", "1000")
    with pkg.renderer.HipPathTracer(0) as pt:
        pt.upload_scene(big)                       # returns at once; its tree is a second away
        pt.upload_scene(small)                     # replaces it
        pt.set_camera(small.camera(96, 54))
        pt.set_settings(small.settings(**kw))
        got = pt.render().copy()
        assert _tree_flags(pkg, pt.last_stats) == 0 and np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert pt.bvh_info().ref_prims == 3        # the Cornell box's three spheres
        pt.upload_scene(big)
    # (the context was destroyed with the second job still running)
    pt = pkg.renderer.HipPathTracer(0)
    pt.upload_scene(big)
    pt.close()


def test_material_coherent_variant_gives_the_same_film(pkg, load_scene):
    """RTX_FUSED_REGROUP launches the variant of the lean flat path tracer whose lanes trade places by vertex class after every
    closest-hit query (an experiment kept for its measurement): another lane computes a path, the path is the same (to the rounding
    of a separately compiled kernel)."""
    sd = load_scene("cornell_box")
    kw = dict(width=200, height=113, samples_per_pixel=40, max_depth=40, seed=12)
    with pkg.renderer.HipPathTracer(0) as pt:
        pt.upload_scene(sd)
        pt.set_camera(sd.camera(kw["width"], kw["height"]))
        pt.set_settings(sd.settings(**kw))
        want = pt.render().copy()
        st0 = pt.last_stats
        for minority in ("1", "16"):
            os.environ["RTX_FUSED_REGROUP"] = minority
            try:
                got = pt.render().copy()
                st = pt.last_stats
            finally:
                os.environ.pop("RTX_FUSED_REGROUP", None)
            # another kernel, compiled on its own: the same arithmetic up to how the compiler contracts and orders it
            bad = (np.abs(got - want) > 1e-3 * (1.0 + np.abs(want))).any(axis=2)
            assert bad.mean() < 0.01, (minority, bad.mean())
            assert abs(float(got.mean()) - float(want.mean())) < 1e-4 * float(want.mean())
            assert st.samples == st0.samples and abs(int(st.segments) - int(st0.segments)) <= 0.003 * st0.segments and abs(int(st.rays_shadow) - int(st0.rays_shadow)) <= 0.003 * st0.rays_shadow
