"""The synthetic BVH stress configuration (10 M triangles, the fifth benchmark configuration) on the GPU, through the C ABI.

(a) closest hits of 1 M incoherent rays and of the 1920x1080 pixel-centre rays: rtx_trace_closest (double precision, reference
    order) and the production walk in double precision must equal the REAL reference's table (hits/c5.npz: digest of every
    (hit, primitive, t bits) record + the first 10 000 records verbatim) bit for bit; the single-precision production walk is
    held to its measured mismatch rate;
(b) a small film of the queue-based pipeline (the one this configuration runs in) against the CPU oracle on the same numbers;
(c) the full 1920x1080 film at reduced spp: sample count, shard identity, mean against the oracle's small film."""
import numpy as np
import pytest

from conftest import record_rate
from test_c5_oracle import check_against_fixture

pytestmark = pytest.mark.gpu

# Measured on MI355X (profiles/r02_parity_rates.jsonl): the single-precision walk picks another primitive than the
# double-precision reference for < 2e-5 of these rays (grazing hits on triangle edges).
F32_ID_MISMATCH_MAX = 1e-4


@pytest.fixture(scope="module")
def c5_tracer(pkg, c5):
    pt = pkg.renderer.HipPathTracer(0)
    pt.upload_scene(c5["scene"])
    yield pt
    pt.close()


def test_c5_bvh_info(c5_tracer):
    info = c5_tracer.bvh_info()
    assert info.ref_prims == 10_000_000 and info.ref_nodes > 10_000_000
    assert 0 < info.wide_nodes < info.ref_nodes and info.wide_bytes >= 64 * info.wide_nodes
    record_rate("c5_bvh", ref_nodes=int(info.ref_nodes), wide_nodes=int(info.wide_nodes), wide_bytes=int(info.wide_bytes), build_seconds=float(info.build_seconds))


@pytest.mark.parametrize("name", ["random", "pixels"])
def test_c5_closest_hits_equal_the_reference(name, pkg, c5, c5_tracer):
    rays = c5["rays"][name]
    ref = c5_tracer.trace_closest(rays)                           # Scene::hit order, double precision
    check_against_fixture(pkg, c5["fx"], name, ref)
    fast = c5_tracer.trace_closest(rays, pkg.abi.PRECISION_F64)   # the production structure, double precision
    assert pkg.synthetic.hit_table_digest(fast) == str(c5["fx"][name + "_digest"])
    f32 = c5_tracer.trace_closest(rays, pkg.abi.PRECISION_F32)    # the production walk
    agree = (f32["hit"] == ref["hit"]) & (f32["prim_type"] == ref["prim_type"]) & (f32["prim_index"] == ref["prim_index"])
    rate = 1.0 - float(agree.mean())
    record_rate("c5_f32_id_mismatch_" + name, rate=rate, rays=int(len(rays)))
    assert rate <= F32_ID_MISMATCH_MAX, rate
    both = agree & (ref["hit"] == 1)
    assert np.abs(f32["t"][both] - ref["t"][both]).max() < 1e-4 * (1.0 + np.abs(ref["t"][both]).max())


def test_c5_wavefront_film_equals_oracle(pkg, oracle, c5, c5_tracer):
    sd = c5["scene"]
    W, H, spp = 64, 36, 4
    over = dict(width=W, height=H, samples_per_pixel=spp, max_depth=8, seed=97)
    osc = oracle.OracleScene(sd)
    ref, ost = osc.render(sd.camera(W, H), sd.settings(**over), oracle.RNG_COUNTER_PCG, threads=16)
    osc.close()
    pt = c5_tracer
    pt.set_camera(sd.camera(W, H))
    pt.set_settings(sd.settings(precision=pkg.abi.PRECISION_F32, **over))   # pipeline AUTO: the queue-based one for a scene that walks a tree
    got = pt.render().astype(np.float64)
    st = pt.last_stats
    assert st.pipeline_used == pkg.abi.PIPELINE_WAVEFRONT and st.samples == W * H * spp
    assert abs(int(st.segments) - int(ost.segments)) <= 0.003 * ost.segments and abs(int(st.rays_shadow) - int(ost.rays_shadow)) <= 0.003 * ost.rays_shadow
    bad = (np.abs(got - ref) > 2e-3 * (1.0 + np.abs(ref))).any(axis=2)
    record_rate("c5_wavefront_film_bad_pixels", rate=float(bad.mean()))
    assert bad.mean() < 0.02, (bad.mean(), np.abs(got - ref).max())
    assert abs(got.mean() - ref.mean()) < 0.01 * (1 + ref.mean())
    # the fused kernel walks the same tree: same film to rounding
    pt.set_settings(sd.settings(precision=pkg.abi.PRECISION_F32, pipeline=pkg.abi.PIPELINE_FUSED, **over))
    fused = pt.render().astype(np.float64)
    assert ((np.abs(fused - got) > 1e-3 * (1.0 + np.abs(got))).any(axis=2)).mean() < 0.01


def test_c5_full_size_properties(pkg, c5, c5_tracer):
    sd = c5["scene"]
    W, H, spp = 1920, 1080, 4
    pt = c5_tracer
    pt.set_camera(sd.camera(W, H))
    base = dict(width=W, height=H, samples_per_pixel=spp, max_depth=8, seed=1234)
    pt.set_settings(sd.settings(**base))
    film = pt.render().copy()
    st = pt.last_stats
    assert st.pipeline_used == pkg.abi.PIPELINE_WAVEFRONT and st.samples == W * H * spp
    assert st.rays_closest > st.samples and st.rays_shadow > 0 and np.isfinite(film).all() and film.mean() > 1e-4   # a dark scene: the soup hides most of the light
    record_rate("c5_full_size", seconds_render=float(st.seconds_render), msamples_per_s=W * H * spp / st.seconds_render / 1e6,
                mrays_per_s=(st.rays_closest + st.rays_shadow) / st.seconds_render / 1e6)
    again = pt.render()
    assert np.array_equal(again.view(np.uint32), film.view(np.uint32))   # reproducible bit for bit
    for k, n in ((5, 8), (1, 2)):
        pt.set_settings(sd.settings(shard_index=k, shard_count=n, **base))
        assert np.array_equal(pt.render().view(np.uint32), film[k::n].view(np.uint32)), (k, n)


def test_c5_first_frame_does_not_wait_for_the_reference_order_tree(pkg, c5):
    """rtx_upload_scene returns before the reference-order tree of a scene this size is built; a frame rendered at once decides
    exact ties in t by primitive number, notes the work items they occur in, and renders those again when the tree is there: the
    film equals the one rendered with the tree, bit for bit."""
    import time
    abi = pkg.abi
    sd = c5["scene"]
    W, H, spp = 1920, 1080, 8
    with pkg.renderer.HipPathTracer(0) as pt:
        t0 = time.perf_counter()
        pt.upload_scene(sd)
        upload_s = time.perf_counter() - t0
        pt.set_camera(sd.camera(W, H))
        pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=spp, max_depth=8, seed=1234))
        first = pt.render().copy()
        f1 = pt.last_stats.flags
        info = pt.bvh_info()                       # waits for the tree (if the frame has not already)
        second = pt.render().copy()
        f2 = pt.last_stats.flags & ~abi.STAT_WALKS_SIDE_BY_SIDE   # (a small frame: the two walks of a turn ran side by side)
    record_rate("c5_upload_before_tree", upload_seconds=upload_s, reference_tree_seconds=float(info.build_seconds), first_frame_flags=int(f1))
    assert f1 & abi.STAT_BEFORE_TREE and not f1 & abi.STAT_RENDERED_TWICE and f2 == 0
    assert np.array_equal(first.view(np.uint32), second.view(np.uint32))
    assert upload_s < info.build_seconds + 0.25    # the call did not include the tree (0.14 s against 0.39 s measured)
