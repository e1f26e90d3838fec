"""Parity tests proper: the HIP path, through the C ABI, against (a) fixtures the real reference
produced and (b) the CPU oracle on the same seeded inputs.  Run with -m gpu on an MI355X."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, record_rate, rel_mse, soup_scene

pytestmark = pytest.mark.gpu


# Single-precision PathPhoton film against the oracle on the same numbers: share of pixels off by more than 2e-3 relative.
# Measured: 0 to 0.0073 (7 of 960 pixels, sphere_field); a photon map with missing or misplaced photons moves most lit pixels.
PATHPHOTON_F32_BAD_PIXELS_MAX = 0.015
# Single-precision production walk against the double-precision reference order on the fixture ray sets (pixel centres +
# random rays with zero components, axis-parallel and short directions): share of rays whose winning primitive differs.
# Measured per scene on MI355X: profiles/r02_parity_rates.jsonl.
# The walk is deterministic; measured counts of such rays (grazing hits on silhouettes and edges): normal_mapped 3, kitchen_sink 1,
# soup 1, every other scene 0, of 4296 (soup: 10096) rays.  The bound leaves room for the two or three grazing rays another
# compiler version may round the other way (5 rays = 0.12 % of a set; a walk that skips a node or a primitive differs in hundreds);
# the measured count itself is recorded with every run (record_rate), so a drift shows.
F32_ID_MISMATCH_RAYS_MAX = 5
# Per-channel mean of the converged film against the reference's, relative: 1 %.
CHANNEL_MEAN_TOL = 0.01
PHOTON_CHANNEL_MEAN_TOL = 0.05   # PathPhoton: the mean of eight films with eight photon maps against the reference film with its ONE map (measured: 0.7 / 2.6 / 3.7 %; one map against one map: 0.7 / 3.4 / 4.7 %)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def tracer(pkg):
    assert pkg.renderer.test_hip(0), "rtx_selftest failed"
    made = []

    def _make(scene):
        pt = pkg.renderer.HipPathTracer(0)
        pt.upload_scene(scene)
        made.append(pt)
        return pt
    yield _make
    for pt in made:
        pt.close()


def check_hits_exact(got, exp):
    for f in ("hit", "prim_type", "prim_index", "material_id", "front_face"):
        assert np.array_equal(got[f], exp[f]), f
    # t, point, normal only involve + - * / sqrt, all correctly rounded on gfx950 with
    # contraction off: every bit must match the CPU renderer
    for f in ("t", "point", "normal"):
        assert np.array_equal(bits(got[f]), bits(exp[f])), f
    # u, v go through acos / atan2, whose device implementations are not the host's
    assert np.abs(got["u"] - exp["u"]).max() < 1e-12 and np.abs(got["v"] - exp["v"]).max() < 1e-12


@pytest.mark.parametrize("name", ["cornell_box", "cornell_box_quad", "environment_demo", "demo", "kitchen_sink", "textured", "disk_room", "normal_mapped"])
def test_closest_hit_bit_exact_vs_reference(name, tracer, load_scene):
    """Primary-ray (pixel-centre) and random ray sets, incl. zero-component and axis-parallel
    directions: primitive ids and t ordering equal Scene::hit's (engine/scene.hpp:574-674)."""
    fx = np.load(os.path.join(GOLDEN, "hits", name + ".npz"))
    pt = tracer(load_scene(name))
    check_hits_exact(pt.trace_closest(fx["rays"]), fx["hits"])


def test_closest_hit_triangle_soup_bit_exact_vs_reference(pkg, tracer, load_scene):
    fx = np.load(os.path.join(GOLDEN, "hits", "soup.npz"))
    pt = tracer(soup_scene(pkg, load_scene))
    info = pt.bvh_info()
    assert info.ref_prims == 4002 and info.ref_nodes > 4000
    check_hits_exact(pt.trace_closest(fx["rays"]), fx["hits"])
    # the production BVH walk (nearest-child-first order) in double: same winners and t bits
    fast = pt.trace_closest(fx["rays"], pkg.abi.PRECISION_F64)
    same = (fast["hit"] == fx["hits"]["hit"]) & (fast["prim_type"] == fx["hits"]["prim_type"]) & (fast["prim_index"] == fx["hits"]["prim_index"])
    assert same.all() and np.array_equal(bits(fast["t"]), bits(fx["hits"]["t"]))
    # single precision: report-level agreement only
    f32 = pt.trace_closest(fx["rays"], pkg.abi.PRECISION_F32)
    agree = (f32["hit"] == fx["hits"]["hit"]) & (f32["prim_type"] == fx["hits"]["prim_type"]) & (f32["prim_index"] == fx["hits"]["prim_index"])
    record_rate("f32_id_mismatch_soup", rate=1.0 - float(agree.mean()), mismatching=int((~agree).sum()), rays=int(len(agree)))
    assert int((~agree).sum()) <= F32_ID_MISMATCH_RAYS_MAX, int((~agree).sum())
    hit = agree & (f32["hit"] == 1)
    assert np.abs(f32["t"][hit] - fx["hits"]["t"][hit]).max() < 1e-3 * (1 + np.abs(fx["hits"]["t"][hit]).max())


@pytest.mark.parametrize("name", ["cornell_box", "cornell_box_quad", "environment_demo", "demo", "kitchen_sink", "textured", "disk_room", "normal_mapped"])
def test_fast_path_closest_hit_agreement(name, pkg, tracer, load_scene):
    fx = np.load(os.path.join(GOLDEN, "hits", name + ".npz"))
    pt = tracer(load_scene(name))
    f64 = pt.trace_closest(fx["rays"], pkg.abi.PRECISION_F64)
    check_hits_exact(f64, fx["hits"])
    f32 = pt.trace_closest(fx["rays"], pkg.abi.PRECISION_F32)
    agree = (f32["hit"] == fx["hits"]["hit"]) & (f32["prim_type"] == fx["hits"]["prim_type"]) & (f32["prim_index"] == fx["hits"]["prim_index"])
    record_rate("f32_id_mismatch_" + name, rate=1.0 - float(agree.mean()), mismatching=int((~agree).sum()), rays=int(len(agree)))
    assert int((~agree).sum()) <= F32_ID_MISMATCH_RAYS_MAX, int((~agree).sum())


@pytest.mark.parametrize("case", ["kitchen_sink", "soup", "stress_300k_with_duplicates", "stress_1500k"])
def test_reference_order_tree_equals_the_oracles(case, pkg, oracle, tracer, load_scene):
    """The reference-order tree the library builds (32-byte keys, node boxes bottom-up, large subtrees on host threads of their
    own) is, node for node and primitive for primitive, the tree BVH::build makes (engine/bvh.hpp:148-218), which the oracle
    restates with the reference's own 56-byte primitives and one std::nth_element per node: boxes bit for bit, child / leaf
    fields, and the order std::nth_element leaves the primitives in - also where centroids are equal."""
    if case in ("kitchen_sink", "soup"):
        sd = soup_scene(pkg, load_scene) if case == "soup" else load_scene(case)
    else:
        n = 300_000 if case.startswith("stress_300k") else 1_500_000
        sd = pkg.synthetic.stress_scene(n, seed=11)
        if case.endswith("duplicates"):
            tri = sd.triangle_array
            tri[1::7] = tri[0:-1:7][: len(tri[1::7])]          # every seventh triangle twice: equal centroids on every axis
            tri["v0"][2::11, 0] = 1.25                          # and runs of equal x
            sd.triangle_array = tri
    pt = tracer(sd)
    boxes, ints, prims = pt.reference_bvh()
    oboxes, oints, oprims = oracle.OracleScene(sd).bvh()
    assert prims.shape == oprims.shape and np.array_equal(prims, oprims)
    assert np.array_equal(ints, oints)
    assert np.array_equal(bits(boxes), bits(oboxes))


@pytest.mark.parametrize("n_tris,n_spheres,n_boxes", [(13, 0, 0), (0, 13, 0), (5, 5, 4), (27, 3, 2), (64, 0, 1), (65, 7, 0), (500, 13, 11), (5000, 50, 50), (40000, 0, 0)])
def test_mixed_primitive_trees_of_many_sizes(n_tris, n_spheres, n_boxes, pkg, tracer, load_scene):
    """Trees of one node, of odd and even numbers of inner children, and of several levels, over triangles, spheres and boxes
    mixed: the production walks find the primitive the reference-order walk (Scene::hit, engine/scene.hpp:574-674) finds.  In
    double every ray must agree, bits of t included; in single precision a ray that grazes an edge may differ."""
    import copy
    sd = copy.copy(load_scene("cornell_box_quad"))
    sd._keep = []
    sd.spheres, sd.boxes, sd.triangles = [], [], []
    seed = 1000 + n_tris + 7 * n_spheres + 13 * n_boxes
    size = 0.5 if n_tris + n_spheres + n_boxes < 100 else 0.15
    if n_tris:
        soup = pkg.synthetic.triangle_soup(n_tris, seed=seed, extent=2.2, size=size, n_materials=3)
        for k in ("v0", "v1", "v2"):
            soup[k] += (0, 2.5, -2.5)
        sd.triangle_array = soup
    else:
        sd.triangle_array = None
    idx = np.arange(max(n_spheres, n_boxes, 1), dtype=np.uint64)
    u = [pkg.synthetic.hash_uniform(idx, 40 + k, seed) for k in range(7)]
    for i in range(n_spheres):
        s = pkg.abi.Sphere()
        s.center[:] = ((u[0][i] * 2 - 1) * 2.2, 2.5 + (u[1][i] * 2 - 1) * 2.2, -2.5 + (u[2][i] * 2 - 1) * 2.2)
        s.radius, s.material_id = size * (0.3 + u[3][i]), i % 3
        sd.spheres.append(s)
    for i in range(n_boxes):
        b = pkg.abi.Box()
        lo = np.array(((u[4][i] * 2 - 1) * 2.2, 2.5 + (u[5][i] * 2 - 1) * 2.2, -2.5 + (u[6][i] * 2 - 1) * 2.2))
        b.box_min[:] = tuple(lo)
        b.box_max[:] = tuple(lo + size * (0.2 + np.array((u[0][i], u[1][i], u[2][i]))))
        b.material_id = i % 3
        sd.boxes.append(b)
    pt = tracer(sd)
    info = pt.bvh_info()
    assert info.ref_prims == n_tris + n_spheres + n_boxes and info.wide_nodes >= 1
    rays = pkg.synthetic.stress_rays(20000, seed=seed, half=2.7)
    rays[:, 1] += 2.5
    rays[:, 2] -= 2.5
    exp = pt.trace_closest(rays)
    assert exp["hit"].mean() > 0.5
    f64 = pt.trace_closest(rays, pkg.abi.PRECISION_F64)
    for f in ("hit", "prim_type", "prim_index"):
        assert np.array_equal(f64[f], exp[f]), f
    assert np.array_equal(bits(f64["t"]), bits(exp["t"]))
    f32 = pt.trace_closest(rays, pkg.abi.PRECISION_F32)
    differ = int(((f32["hit"] != exp["hit"]) | (f32["prim_type"] != exp["prim_type"]) | (f32["prim_index"] != exp["prim_index"])).sum())
    record_rate(f"f32_id_mismatch_mixed_{n_tris}_{n_spheres}_{n_boxes}", rate=differ / len(rays), mismatching=differ, rays=len(rays))
    assert differ <= 2, differ   # measured: 0 of 20 000 in every case
    hit = (f32["hit"] == 1) & (exp["hit"] == 1) & (f32["prim_index"] == exp["prim_index"]) & (f32["prim_type"] == exp["prim_type"])
    assert np.abs(f32["t"][hit] - exp["t"][hit]).max() < 1e-3 * (1 + np.abs(exp["t"][hit]).max())


def test_edge_cases_empty_and_degenerate(pkg, tracer, load_scene):
    pt = tracer(load_scene("cornell_box"))
    assert pt.trace_closest(np.zeros((0, 8))).shape == (0,)
    rays = np.array([[0, 2.5, 2, 0, 0, -1, 0.001, 1e9],      # straight at the back wall
                     [0, 2.5, 2, 0, 0, -1, 0.001, 0.5],      # t_max before anything
                     [0, 2.5, 2, 0, 0, 0, 0.001, 1e9],       # zero direction: no hit, no fault
                     [0, 2.5, 2, 0, 1, 0, 2.5, 2.5]])        # t_min == t_max == the ceiling's t: accepted (ties count)
    h = pt.trace_closest(rays)
    assert h["hit"].tolist() == [1, 0, 0, 1] and h["t"][0] == 7.0 and h["t"][3] == 2.5 and h["prim_type"][0] == 1


@pytest.mark.parametrize("name,nee,mis", [("cornell_box", 1, 1), ("cornell_box", 1, 0), ("cornell_box", 0, 0), ("cornell_box_quad", 1, 1),
                                           ("caustics_test", 1, 1), ("environment_demo", 1, 1), ("environment_demo", 1, 0), ("demo", 1, 1),
                                           ("kitchen_sink", 1, 1), ("kitchen_sink", 0, 1), ("textured", 1, 1), ("disk_room", 1, 1), ("earth_demo", 1, 1),
                                           ("normal_mapped", 1, 1), ("normal_mapped", 1, 0)])
def test_f64_film_equals_oracle(name, nee, mis, pkg, oracle, tracer, load_scene):
    """Double-precision render vs the CPU oracle driven by the same counter-based sampler: the
    two run the same arithmetic on the same random numbers, so films agree to rounding of the
    transcendental functions.  A path may flip a branch on such a rounding difference; those
    pixels are bounded to a 0.2 % fraction."""
    sd = load_scene(name)
    W, H, spp = 48, 27, 8
    over = dict(width=W, height=H, samples_per_pixel=spp, max_depth=min(sd.max_depth, 40), seed=31, use_nee=nee, use_mis=mis)
    ref, ost = oracle.OracleScene(sd).render(sd.camera(W, H), sd.settings(**over), oracle.RNG_COUNTER, threads=16)
    pt = tracer(sd)
    pt.set_camera(sd.camera(W, H))
    pt.set_settings(sd.settings(precision=pkg.abi.PRECISION_F64, **over))
    got = pt.render().astype(np.float64)
    st = pt.last_stats
    assert st.samples == W * H * spp and st.rays_shadow == ost.rays_shadow and st.segments == ost.segments
    bad = np.abs(got - ref) > 1e-5 * (1.0 + np.abs(ref))
    # kitchen_sink's checker floor lies in the plane y = 0: hit points have y = +-1e-17 and
    # floor(y / scale) is -1 or 0 on the sign of that residue (texture.hpp:276-280), so the
    # checker colour there flips on last-bit differences of sin/cos — in the reference too.
    limit, mean_tol = (0.03, 3e-3) if name == "kitchen_sink" else (0.002, 1e-4)
    assert bad.mean() < limit, (bad.mean(), np.abs(got - ref).max())
    assert abs(got.mean() - ref.mean()) < mean_tol * (1 + ref.mean())


@pytest.mark.parametrize("name,pipe", [("cornell_box", "fused"), ("cornell_box_quad", "fused"), ("environment_demo", "fused"), ("textured", "fused"),
                                       ("cornell_box", "wavefront"), ("soup", "wavefront"), ("normal_mapped", "fused"), ("normal_mapped", "wavefront"),
                                       ("sphere_field", "fused"), ("sphere_field", "wavefront"), ("earth_demo", "fused")])
def test_f32_pcg_film_equals_oracle_pcg(name, pipe, pkg, oracle, tracer, load_scene):
    """Single precision draws its samples from the pcg4d block generator (RTX_RNG_AUTO).  The oracle fed by the same
    generator renders the same paths in double precision: films agree to single-precision rounding, apart from the
    rare sample whose path flips on a rounding difference."""
    sd = soup_scene(pkg, load_scene) if name == "soup" else load_scene(name)
    W, H, spp = 48, 27, 8
    over = dict(width=W, height=H, samples_per_pixel=spp, max_depth=min(sd.max_depth, 30), seed=61)
    ref, ost = oracle.OracleScene(sd).render(sd.camera(W, H), sd.settings(**over), oracle.RNG_COUNTER_PCG, threads=16)
    pt = tracer(sd)
    pt.set_camera(sd.camera(W, H))
    pt.set_settings(sd.settings(precision=pkg.abi.PRECISION_F32, pipeline=pkg.abi.PIPELINE_WAVEFRONT if pipe == "wavefront" else pkg.abi.PIPELINE_FUSED, **over))
    got = pt.render().astype(np.float64)
    st = pt.last_stats
    assert st.samples == W * H * spp
    assert abs(int(st.segments) - int(ost.segments)) <= 0.003 * ost.segments and abs(int(st.rays_shadow) - int(ost.rays_shadow)) <= 0.003 * ost.rays_shadow
    bad = (np.abs(got - ref) > 2e-3 * (1.0 + np.abs(ref))).any(axis=2)
    record_rate("f32_pcg_film_bad_pixels_" + name + "_" + pipe, rate=float(bad.mean()), mean_rel_diff=float(abs(got.mean() - ref.mean()) / (1 + ref.mean())))
    # earth_demo (the image-textured instance of the fused kernel, TREE_UV): its checker floor lies in the plane y = 0, where the
    # checker colour flips on the sign of a rounding residue (see test_f64_film_equals_oracle)
    limit = 0.08 if name == "earth_demo" else 0.02
    assert bad.mean() < limit, (bad.mean(), np.abs(got - ref).max())
    assert abs(got.mean() - ref.mean()) < 0.01 * (1 + ref.mean())
    # and the Philox stream is a different one
    pt.set_settings(sd.settings(precision=pkg.abi.PRECISION_F32, rng=pkg.abi.RNG_PHILOX, pipeline=pkg.abi.PIPELINE_FUSED, **over))
    other = pt.render().astype(np.float64)
    if name != "earth_demo":   # (one point light, mirror-less geometry collapsed onto the origin: most of that film does not depend on the random numbers)
        assert (np.abs(other - got) > 1e-3).mean() > 0.2
    else:
        assert not np.array_equal(other, got)


@pytest.mark.parametrize("name", ["demo", "kitchen_points", "textured", "earth_demo", "kitchen_sink", "normal_mapped"])
def test_whitted_film_equals_oracle(name, pkg, oracle, tracer, load_scene):
    """RenderMode::Whitted (Renderer::ray_color_whitted, engine/renderer.cpp:483-569; the native mode of demo.json and
    earth_demo.json): double precision agrees with the oracle on the same counter-based numbers (the oracle equals the
    real reference bit for bit, test_seeded_film_bit_exact); single precision agrees with double to rounding."""
    sd = load_scene(name)
    W, H, spp = 48, 27, 8
    over = dict(width=W, height=H, samples_per_pixel=spp, max_depth=min(sd.max_depth, 20), seed=53, mode=pkg.abi.MODE_WHITTED)
    ref, ost = oracle.OracleScene(sd).render(sd.camera(W, H), sd.settings(**over), oracle.RNG_COUNTER, threads=16)
    pt = tracer(sd)
    pt.set_camera(sd.camera(W, H))
    pt.set_settings(sd.settings(precision=pkg.abi.PRECISION_F64, **over))
    got = pt.render().astype(np.float64)
    st = pt.last_stats
    loose = name in ("kitchen_sink", "kitchen_points", "earth_demo")   # checker floor in the plane y = 0 (see test_f64_film_equals_oracle)
    if not loose:
        assert st.samples == W * H * spp and st.rays_shadow == ost.rays_shadow and st.segments == ost.segments and st.rays_closest == ost.rays_closest
    bad = (np.abs(got - ref) > 1e-5 * (1.0 + np.abs(ref))).any(axis=2)
    assert bad.mean() < (0.03 if loose else 0.002), (bad.mean(), np.abs(got - ref).max())
    assert abs(got.mean() - ref.mean()) < (3e-3 if loose else 1e-4) * (1 + ref.mean())
    pt.set_settings(sd.settings(precision=pkg.abi.PRECISION_F32, rng=pkg.abi.RNG_PHILOX, **over))   # same numbers as the double run
    f32 = pt.render().astype(np.float64)
    assert pt.last_stats.pipeline_used == pkg.abi.PIPELINE_FUSED and pt.last_stats.samples == W * H * spp
    if name in ("demo", "textured", "normal_mapped"):
        # the other scenes have a checker texture on the plane y = 0: floor(y / scale) there is -1 or 0 on the sign of a
        # rounding residue, which differs between precisions over the whole plane (texture.hpp:276-280)
        bad = (np.abs(f32 - got) > 1e-3 * (1.0 + np.abs(got))).any(axis=2)
        assert bad.mean() < 0.01, (bad.mean(), np.abs(f32 - got).max())
        assert abs(f32.mean() - got.mean()) < 5e-3 * (1 + got.mean())
    else:
        assert abs(f32.mean() - got.mean()) < 0.05 * (1 + got.mean())
    if name == "demo":
        assert got.mean() > 0.05


@pytest.mark.parametrize("name,mis", [("caustics_test", 1), ("cornell_box", 1), ("cornell_box", 0), ("cornell_box_quad", 1), ("kitchen_sink", 1), ("disk_room", 1),
                                      ("textured", 1), ("kitchen_points", 1), ("soup", 1), ("sphere_field", 1)])
def test_bdpt_film_equals_oracle(name, mis, pkg, oracle, tracer, load_scene):
    """RenderMode::BDPT (BDPTIntegrator::Li, engine/bdpt.cpp:71-128; the native mode of caustics_test.json): double precision
    agrees with the oracle on the same counter-based numbers (the oracle equals the real reference bit for bit on 8 seeded
    films, test_seeded_film_bit_exact), single precision with the oracle fed by the pcg4d generator to rounding.  The
    subpaths are the reference's vertex for vertex (equal closest-hit and vertex counters in double precision); visibility
    rays are fewer, because the kernel casts one only for a connection whose BSDF and geometry terms are not zero."""
    sd = soup_scene(pkg, load_scene) if name == "soup" else load_scene(name)
    W, H, spp = 40, 24, 6
    over = dict(width=W, height=H, samples_per_pixel=spp, max_depth=min(sd.max_depth, 12), seed=71, use_mis=mis, mode=pkg.abi.MODE_BDPT)
    ref, ost = oracle.OracleScene(sd).render(sd.camera(W, H), sd.settings(**over), oracle.RNG_COUNTER, threads=16)
    pt = tracer(sd)
    pt.set_camera(sd.camera(W, H))
    pt.set_settings(sd.settings(precision=pkg.abi.PRECISION_F64, **over))
    got = pt.render().astype(np.float64)
    st = pt.last_stats
    loose = name in ("kitchen_sink", "kitchen_points")   # checker floor in the plane y = 0 (see test_f64_film_equals_oracle)
    assert st.samples == W * H * spp and st.pipeline_used == pkg.abi.PIPELINE_FUSED and st.precision_used == pkg.abi.PRECISION_F64
    if not loose:
        assert st.rays_closest == ost.rays_closest and st.segments == ost.segments and 0 < st.rays_shadow <= ost.rays_shadow
    bad = (np.abs(got - ref) > 1e-5 * (1.0 + np.abs(ref))).any(axis=2)
    assert bad.mean() < (0.03 if loose else 0.005), (bad.mean(), np.abs(got - ref).max())
    assert abs(got.mean() - ref.mean()) < (3e-3 if loose else 2e-4) * (1 + ref.mean())
    assert ref.mean() > 1e-3
    # single precision, pcg4d numbers
    ref32, _ = oracle.OracleScene(sd).render(sd.camera(W, H), sd.settings(**over), oracle.RNG_COUNTER_PCG, threads=16)
    pt.set_settings(sd.settings(precision=pkg.abi.PRECISION_F32, **over))
    f32 = pt.render().astype(np.float64)
    assert pt.last_stats.precision_used == pkg.abi.PRECISION_F32 and pt.last_stats.samples == W * H * spp
    if loose:
        # the checker texture on the plane y = 0 takes floor(y / scale) of a rounding residue, whose sign differs between
        # the precisions over the whole plane (texture.hpp:276-280): only the image mean is comparable
        assert abs(f32.mean() - ref32.mean()) < 0.05 * (1 + ref32.mean())
    else:
        bad = (np.abs(f32 - ref32) > 2e-3 * (1.0 + np.abs(ref32))).any(axis=2)
        assert bad.mean() < 0.03, (bad.mean(), np.abs(f32 - ref32).max())
        assert abs(f32.mean() - ref32.mean()) < 0.02 * (1 + ref32.mean())
    # the queue pipeline has no BDPT
    pt.set_settings(sd.settings(pipeline=pkg.abi.PIPELINE_WAVEFRONT, **over))
    with pytest.raises(pkg.abi.RtxError):
        pt.render()


@pytest.mark.parametrize("name,nee,photons,k,radius", [("cornell_box", 1, 4000, 40, 0.3), ("cornell_box", 0, 1500, 10, 0.1), ("kitchen_points", 1, 3000, 25, 0.25),
                                                        ("textured", 1, 3000, 100, 0.5), ("normal_mapped", 1, 3000, 30, 0.3), ("cornell_box_quad", 1, 2000, 20, 0.2),
                                                        ("cornell_box", 1, 60000, 50, 0.15),   # enough photons for a kd-tree several levels deep above the GPU's buckets
                                                        ("cornell_box", 1, 90000, 40, 0.12),   # not walked, and more bounding boxes than the GPU keeps in LDS
                                                        ("cornell_box", 1, 4000, 300, 0.5),     # more photons per estimate than a wave's LDS list holds: lists in device memory
                                                        ("cornell_box", 1, 2500, 100000, 0.45), # more than there are photons: every photon in reach is kept
                                                        ("cornell_box", 1, 40, 5, 0.6), ("cornell_box", 1, 3, 2, 0.6),   # a handful of photons or none: the smallest trees the device build makes
                                                        ("soup_box", 1, 3000, 30, 0.3),         # cornell_box + 600 triangles: the instances that walk a BVH with an LDS stack
                                                        ("sphere_field", 1, 3000, 30, 0.3)])    # 14 spheres: flat render kernel, but the photon pass walks the tree (LDS stack sized from the scene)
def test_pathphoton_film_equals_oracle(name, nee, photons, k, radius, pkg, oracle, tracer, load_scene, monkeypatch):
    """RenderMode::PathPhoton (engine/renderer.cpp:344-372 photon pass, :844-976 integrator): the GPU traces the caustic photons
    (one lane each, counter-based numbers), the host builds the reference's kd-tree over them, the render gathers the k nearest
    at every Lambertian vertex.  Double precision agrees with the oracle run on the same numbers (the oracle equals the real
    reference bit for bit on 5 seeded films); single precision traces slightly different photons, so it is held to the image mean.
    cornell_box_quad has no emissive sphere: no photons, the mode degenerates to its plain path tracer."""
    sd = soup_scene(pkg, load_scene, base="cornell_box", n=600) if name == "soup_box" else load_scene(name)
    # The GPU walks the kd-tree only above subtrees of 64^3 photons and searches those through bounding boxes; the library reads
    # a smaller subtree size from the environment at every render so that maps of test size are walked too.
    if photons == 60000:
        monkeypatch.setenv("RTX_PHOTON_BUCKET", "200")
    elif photons == 1500:
        monkeypatch.setenv("RTX_PHOTON_BUCKET", "1")
    W, H, spp = 40, 24, 4
    over = dict(width=W, height=H, samples_per_pixel=spp, max_depth=min(sd.max_depth, 10), seed=83, use_nee=nee, mode=pkg.abi.MODE_PATHPHOTON,
                caustic_photon_count=photons, photon_gather_count=k, caustic_gather_radius=radius)
    ref, ost = oracle.OracleScene(sd).render(sd.camera(W, H), sd.settings(**over), oracle.RNG_COUNTER, threads=16)
    pt = tracer(sd)
    pt.set_camera(sd.camera(W, H))
    pt.set_settings(sd.settings(precision=pkg.abi.PRECISION_F64, **over))
    got = pt.render().astype(np.float64)
    st = pt.last_stats
    loose = name == "kitchen_points"   # checker floor in the plane y = 0 (see test_f64_film_equals_oracle)
    assert st.samples == W * H * spp and st.pipeline_used == pkg.abi.PIPELINE_FUSED and st.precision_used == pkg.abi.PRECISION_F64
    if not loose:
        assert st.rays_closest == ost.rays_closest and st.rays_shadow == ost.rays_shadow and st.segments == ost.segments
    bad = (np.abs(got - ref) > 1e-5 * (1.0 + np.abs(ref))).any(axis=2)
    assert bad.mean() < (0.03 if loose else 0.005), (bad.mean(), np.abs(got - ref).max())
    assert abs(got.mean() - ref.mean()) < (3e-3 if loose else 2e-4) * (1 + ref.mean())
    if name != "cornell_box_quad":   # the photons do contribute
        plain, _ = oracle.OracleScene(sd).render(sd.camera(W, H), sd.settings(**dict(over, caustic_photon_count=0)), oracle.RNG_COUNTER, threads=16)
        assert np.abs(ref - plain).max() > 1e-3
    ref32, ost32 = oracle.OracleScene(sd).render(sd.camera(W, H), sd.settings(**over), oracle.RNG_COUNTER_PCG, threads=16)
    pt.set_settings(sd.settings(precision=pkg.abi.PRECISION_F32, **over))
    f32 = pt.render().astype(np.float64)
    s32 = pt.last_stats
    assert s32.precision_used == pkg.abi.PRECISION_F32 and s32.samples == W * H * spp
    assert abs(f32.mean() - ref32.mean()) < 0.05 * (1 + ref32.mean()), (f32.mean(), ref32.mean())
    if not loose:
        # per pixel, against the oracle on the same pcg4d numbers: a wrong or incomplete photon map moves most lit pixels,
        # a rounding flip of one photon's path or of a k-th neighbour only a few
        assert abs(int(s32.segments) - int(ost32.segments)) <= 0.003 * ost32.segments and abs(int(s32.rays_shadow) - int(ost32.rays_shadow)) <= 0.003 * ost32.rays_shadow
        assert abs(int(s32.rays_closest) - int(ost32.rays_closest)) <= 0.003 * ost32.rays_closest
        bad = (np.abs(f32 - ref32) > 2e-3 * (1.0 + np.abs(ref32))).any(axis=2)
        record_rate("pathphoton_f32_bad_pixels_" + name + f"_{photons}", rate=float(bad.mean()))
        assert bad.mean() < PATHPHOTON_F32_BAD_PIXELS_MAX, (bad.mean(), np.abs(f32 - ref32).max())
    # the same render again: the context reuses its photon map (nothing to trace or build) and returns the same film and counts
    again = pt.render().astype(np.float64)
    s_again = pt.last_stats
    assert np.array_equal(again, f32) and s_again.rays_closest == s32.rays_closest and s_again.seconds_photon_pass <= s32.seconds_photon_pass
    pt.set_settings(sd.settings(photon_gather_count=0, **{k_: v for k_, v in over.items() if k_ != "photon_gather_count"}))
    with pytest.raises(pkg.abi.RtxError, match="photon_gather_count"):
        pt.render()


@pytest.mark.parametrize("name,mis", [("cornell_box", 1), ("cornell_box", 0), ("cornell_box_quad", 1), ("caustics_test", 1), ("kitchen_sink", 0), ("disk_room", 1), ("normal_mapped", 1)])
def test_mnee_film_equals_oracle(name, mis, pkg, oracle, tracer, load_scene):
    """use_mnee = 1 (the reference's CLI default; MNEE::evaluate, engine/spectral/mnee.hpp:632-763) in double precision:
    the fused kernel agrees with the oracle run on the same counter-based random numbers.  The oracle's MNEE equals the
    real reference's bit for bit (test_seeded_film_bit_exact).  A Newton solve may stop one iteration earlier or later on
    a last-bit difference; those pixels are bounded."""
    sd = load_scene(name)
    W, H, spp = 40, 24, 6
    over = dict(width=W, height=H, samples_per_pixel=spp, max_depth=min(sd.max_depth, 12), seed=47, use_nee=1, use_mis=mis, use_mnee=1)
    ref, ost = oracle.OracleScene(sd).render(sd.camera(W, H), sd.settings(**over), oracle.RNG_COUNTER, threads=16)
    off, _ = oracle.OracleScene(sd).render(sd.camera(W, H), sd.settings(**dict(over, use_mnee=0)), oracle.RNG_COUNTER, threads=16)
    pt = tracer(sd)
    pt.set_camera(sd.camera(W, H))
    pt.set_settings(sd.settings(precision=pkg.abi.PRECISION_F64, **over))
    got = pt.render().astype(np.float64)
    st = pt.last_stats
    assert st.precision_used == pkg.abi.PRECISION_F64 and st.pipeline_used == pkg.abi.PIPELINE_FUSED
    if name == "kitchen_sink":   # its checker floor flips a sample now and then (see test_f64_film_equals_oracle)
        assert st.samples == W * H * spp and abs(int(st.rays_shadow) - int(ost.rays_shadow)) <= 0.005 * ost.rays_shadow
    else:
        assert st.samples == W * H * spp and st.rays_shadow == ost.rays_shadow and st.segments == ost.segments
    # (closest-hit counts differ by construction: the oracle traces the MIS look-ahead ray twice, as the reference does)
    assert abs(int(st.rays_closest) - int(ost.rays_closest)) <= 0.05 * ost.rays_closest
    bad = (np.abs(got - ref) > 1e-5 * (1.0 + np.abs(ref))).any(axis=2)
    limit = 0.03 if name == "kitchen_sink" else 0.01
    assert bad.mean() < limit, (bad.mean(), np.abs(got - ref).max())
    assert abs(got.mean() - ref.mean()) < 2e-3 * (1 + ref.mean())
    if name in ("cornell_box", "caustics_test"):
        assert np.abs(ref - off).max() > 1e-3   # MNEE does contribute in these scenes


@pytest.mark.parametrize("name", ["cornell_box", "caustics_test", "cornell_box_quad", "disk_room"])   # (not kitchen_sink: its checker floor flips samples between any two single-precision kernel instances)
def test_mnee_f32_follows_the_double_precision_walk(name, pkg, tracer, load_scene):
    """use_mnee = 1 in single precision: single-precision closest-hit queries around the double-precision manifold walk.
    On the same (Philox) random numbers it must find the connections the double-precision kernel finds: the pixels that
    receive an MNEE contribution (film with MNEE minus film without, same precision) are the same set to within 3 %
    (measured: 4.76 % of the pixels against 4.74 % on cornell_box, 7.08 % against 7.07 % on caustics_test; before the hits
    on spheres and planes were recomputed in double inside the walk it found a quarter of them), and the films agree per pixel
    as well as they do without MNEE."""
    abi = pkg.abi
    sd = load_scene(name)
    W, H, spp = 96, 64, 32
    pt = tracer(sd)
    pt.set_camera(sd.camera(W, H))
    over = dict(width=W, height=H, samples_per_pixel=spp, max_depth=min(sd.max_depth, 12), seed=47, use_nee=1, use_mis=1, rng=abi.RNG_PHILOX,
                pipeline=abi.PIPELINE_FUSED)   # all four films from the fused kernels: the queue pipeline rounds differently
    films = {}
    for tag, kw in (("f64_on", dict(precision=abi.PRECISION_F64, use_mnee=1)), ("f32_on", dict(precision=abi.PRECISION_F32, use_mnee=1)),
                    ("f64_off", dict(precision=abi.PRECISION_F64, use_mnee=0)), ("f32_off", dict(precision=abi.PRECISION_F32, use_mnee=0))):
        pt.set_settings(sd.settings(**over, **kw))
        films[tag] = pt.render().astype(np.float64).copy()
        if tag == "f32_on":
            st = pt.last_stats
            assert st.precision_used == abi.PRECISION_F32 and st.pipeline_used == abi.PIPELINE_FUSED and st.samples == W * H * spp
    assert np.isfinite(films["f32_on"]).all()
    with64 = (films["f64_on"] - films["f64_off"]).max(axis=2) > 1e-4
    with32 = (films["f32_on"] - films["f32_off"]).max(axis=2) > 1e-4
    record_rate("mnee_pixels_" + name, f64=float(with64.mean()), f32=float(with32.mean()), both=float((with64 & with32).mean()))
    if name in ("cornell_box", "caustics_test"):
        assert with64.mean() > 0.02                                   # MNEE does contribute in these scenes
    assert (with64 ^ with32).sum() <= 0.03 * max(1, with64.sum()) + 2
    def bad(a, b): return float((np.abs(a - b) > 2e-3 * (1.0 + np.abs(b))).any(axis=2).mean())
    assert bad(films["f32_on"], films["f64_on"]) <= bad(films["f32_off"], films["f64_off"]) + 0.002
    part64, part32 = (films["f64_on"] - films["f64_off"]).mean(), (films["f32_on"] - films["f32_off"]).mean()
    assert abs(part32 - part64) <= 0.15 * abs(part64) + 1e-7   # a handful of bright connections carry the mean (measured: 5 % apart on cornell_box)


@pytest.mark.parametrize("name", ["cornell_box", "caustics_test", "textured", "disk_room"])
def test_mnee_f32_batched_solves_equal_solves_on_the_spot(name, pkg, load_scene, monkeypatch):
    """Single-precision manifold NEE launches the kernel that batches the solves of a wave (a vertex notes its solve, the wave
    solves when most lanes hold one).  Against the instance that solves at the vertex (RTX_MNEE_ON_THE_SPOT=1): the same solves,
    so the same counters; their contributions are summed apart from the paths', so the films agree to rounding.  And the film must
    not depend on the grid (a contribution is a function of its task alone): bit-equal with another launch geometry."""
    abi = pkg.abi
    sd = load_scene(name)
    W, H, spp = 96, 64, 24
    kw = dict(width=W, height=H, samples_per_pixel=spp, max_depth=min(sd.max_depth, 12), seed=5, use_nee=1, use_mis=1, use_mnee=1, precision=abi.PRECISION_F32,
              pipeline=abi.PIPELINE_FUSED)

    def render(env):
        for k in ("RTX_MNEE_ON_THE_SPOT", "RTX_ITEMS_PER_LANE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with pkg.renderer.HipPathTracer(0) as pt:
            pt.upload_scene(sd)
            pt.set_camera(sd.camera(W, H))
            pt.set_settings(sd.settings(**kw))
            film = pt.render().copy()
            return film, pt.last_stats

    batched, sb = render({})
    spot, ss = render({"RTX_MNEE_ON_THE_SPOT": "1"})
    assert (sb.samples, sb.rays_shadow, sb.segments) == (ss.samples, ss.rays_shadow, ss.segments)
    # (the solver is compiled twice, once into each kernel, and its double-precision algebra is contracted differently: a solve in a
    # hundred thousand takes one probe more or fewer, measured 121 of 31.6 M queries on cornell_box)
    assert abs(int(sb.rays_closest) - int(ss.rays_closest)) <= 1e-4 * ss.rays_closest
    assert np.isfinite(batched).all()
    differs = np.abs(batched.astype(np.float64) - spot).max(axis=2) > 1e-4 * (1.0 + np.abs(spot).max())
    assert differs.mean() <= 0.001
    other_grid, so = render({"RTX_ITEMS_PER_LANE": "2"})
    assert np.array_equal(other_grid.view(np.uint32), batched.view(np.uint32)) and so.rays_closest == sb.rays_closest
    half = dict(kw, shard_index=1, shard_count=2)
    with pkg.renderer.HipPathTracer(0) as pt:
        pt.upload_scene(sd)
        pt.set_camera(sd.camera(W, H))
        pt.set_settings(sd.settings(**half))
        shard = pt.render().copy()
    assert np.array_equal(shard.view(np.uint32), batched[1::2].view(np.uint32))


@pytest.mark.parametrize("name", ["cornell_box", "cornell_box_quad", "caustics_test", "environment_demo", "caustics_test_bdpt", "cornell_box_quad_bdpt",
                                  "cornell_box_pathphoton"])
def test_f32_film_within_tolerance_of_reference(name, pkg, tracer, load_scene):
    """Converged-image criterion: relMSE(GPU @ N spp vs reference @ 4N spp) <= 1.25 x
    relMSE(reference @ N vs reference @ 4N, other seeds), relMSE = mean((x-r)^2 / (r^2 + 0.01)) on
    the clamped-linear film, and per-channel means within 1 % (measured: at most 0.45 %, profiles/r02_parity_rates.jsonl).  relMSE of one N-spp film scatters by 12-20 % from
    seed to seed (scripts/rng_quality.py, 64 seeds, either generator), so it is averaged over four seeds here; the
    means are taken from a 16N-spp film, whose own noise is below the reference film's."""
    fx = np.load(os.path.join(GOLDEN, "converged", name + ".npz"))
    sd = load_scene(name.rsplit("_", 1)[0] if name.endswith(("_bdpt", "_pathphoton")) else name)   # the reference's films in those render modes
    mode = int(fx["mode"]) if "mode" in fx else pkg.abi.MODE_PATHTRACE
    extra = {}
    if "caustic_photon_count" in fx:   # PathPhoton: every film has its own photon map, whose variance is part of the floor and of the means
        extra = dict(caustic_photon_count=int(fx["caustic_photon_count"]), photon_gather_count=int(fx["photon_gather_count"]), caustic_gather_radius=float(fx["caustic_gather_radius"]))
    W, H, N = int(fx["W"]), int(fx["H"]), int(fx["N"])
    pt = tracer(sd)
    pt.set_camera(sd.camera(W, H))
    ref = fx["film"].astype(np.float64)
    errs = []
    for seed in (2024, 2026, 2027, 2028):
        pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=N, max_depth=int(fx["depth"]), seed=seed, mode=mode, **extra))
        errs.append(rel_mse(pt.render().astype(np.float64), ref))
    assert np.mean(errs) <= 1.25 * float(fx["noise_floor_relmse"]), (errs, float(fx["noise_floor_relmse"]))
    assert max(errs) <= 1.8 * float(fx["noise_floor_relmse"]), errs
    # many more samples than the reference: error must drop to the reference film's own noise
    if extra:
        # A PathPhoton film keeps the variance of its one photon map however many samples it takes: the mean of films with
        # different maps (the map follows the seed) does not.  What is left is the variance of the REFERENCE film's one map.
        films = []
        for seed in range(2025, 2025 + 8):
            pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=2 * N, max_depth=int(fx["depth"]), seed=seed, mode=mode, **extra))
            films.append(pt.render().astype(np.float64))
        film16 = np.mean(films, axis=0)
    else:
        pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=16 * N, max_depth=int(fx["depth"]), seed=2025, mode=mode, **extra))
        film16 = pt.render().astype(np.float64)
    assert rel_mse(film16, ref) <= (0.85 if extra else 0.6) * float(fx["noise_floor_relmse"])   # (PathPhoton, measured: 0.75 of the floor)
    mg, mr = np.clip(film16, 0, 1).mean(axis=(0, 1)), np.clip(ref, 0, 1).mean(axis=(0, 1))
    record_rate("converged_" + name, relmse_mean=float(np.mean(errs)), relmse_max=float(max(errs)), noise_floor=float(fx["noise_floor_relmse"]),
                relmse_16n=float(rel_mse(film16, ref)), channel_mean_rel_diff=[float(v) for v in (mg - mr) / mr])
    assert np.all(np.abs(mg - mr) / mr < (PHOTON_CHANNEL_MEAN_TOL if extra else CHANNEL_MEAN_TOL)), (mg, mr)


@pytest.mark.parametrize("mode", ["pathtrace", "whitted", "bdpt", "pathphoton"])
def test_render_is_reproducible_and_shard_invariant(mode, pkg, tracer, load_scene):
    """Counter-based sampling keyed on the absolute pixel: same seed => same film, bit for bit;
    the union of row shards (any count) equals the unsharded film, bit for bit — in every render mode (every shard of a
    PathPhoton render traces the same photon map)."""
    sd = load_scene("kitchen_points" if mode == "whitted" else "cornell_box")   # Whitted lights from point lights only
    W, H = 96, 54
    pt = tracer(sd)
    pt.set_camera(sd.camera(W, H))
    extra = {"pathtrace": {}, "whitted": dict(mode=pkg.abi.MODE_WHITTED), "bdpt": dict(mode=pkg.abi.MODE_BDPT),
             "pathphoton": dict(mode=pkg.abi.MODE_PATHPHOTON, caustic_photon_count=5000, photon_gather_count=20, caustic_gather_radius=0.2)}[mode]
    base = dict(width=W, height=H, samples_per_pixel=24 if mode == "pathtrace" else 6, max_depth=30 if mode == "pathtrace" else 10, seed=9, **extra)
    pt.set_settings(sd.settings(**base))
    a = pt.render().copy()
    b = pt.render().copy()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    pt.set_settings(sd.settings(**dict(base, seed=10)))
    assert not np.array_equal(a, pt.render())
    for n in (2, 3, 8):
        full = np.zeros_like(a)
        for k in range(n):
            pt.set_settings(sd.settings(shard_index=k, shard_count=n, **base))
            part = pt.render()
            assert part.shape[0] == len(range(k, H, n))
            full[k::n] = part
        assert np.array_equal(full.view(np.uint32), a.view(np.uint32)), n


def test_render_device_buffer_matches_host_buffer(pkg, tracer, load_scene):
    import torch
    sd = load_scene("cornell_box_quad")
    W, H = 64, 64
    pt = tracer(sd)
    pt.set_camera(sd.camera(W, H))
    pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=8, max_depth=10, seed=3))
    host = pt.render().copy()
    dev = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    st = pt.render_device(dev.data_ptr())
    assert st.samples == W * H * 8
    assert np.array_equal(dev.cpu().numpy(), host)


def test_full_size_properties(pkg, tracer, load_scene):
    """The headline configuration (Cornell box, 1920x1080, NEE+MIS, depth 500) at reduced
    spp: sample count, film mean against the reference's converged low-resolution film, and
    agreement between two shards and the full film on their rows."""
    fx = np.load(os.path.join(GOLDEN, "converged", "cornell_box.npz"))
    sd = load_scene("cornell_box")
    W, H, spp = 1920, 1080, 32
    pt = tracer(sd)
    pt.set_camera(sd.camera(W, H))
    pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=spp, max_depth=500, seed=1234))
    film = pt.render()
    st = pt.last_stats
    assert st.samples == W * H * spp and st.rays_closest > st.samples and st.rays_shadow > 0
    # The mean of the UNCLAMPED film estimates the same image-plane integral at any resolution
    # (clamped means do not: a 64x36 pixel averages highlights before the clamp).
    mg = film.astype(np.float64).mean(axis=(0, 1))
    mr = fx["film"].astype(np.float64).mean(axis=(0, 1))
    record_rate("full_size_cornell_box_unclamped_mean", channel_mean_rel_diff=[float(v) for v in (mg - mr) / mr])
    assert np.all(np.abs(mg - mr) / mr < CHANNEL_MEAN_TOL), (mg, mr)
    pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=spp, max_depth=500, seed=1234, shard_index=5, shard_count=8))
    part = pt.render()
    assert np.array_equal(part, film[5::8])


def test_error_paths(pkg, load_scene):
    import ctypes as C
    abi = pkg.abi
    sd = load_scene("cornell_box")
    with pkg.renderer.HipPathTracer(0) as pt:
        pt.set_settings(sd.settings(width=8, height=8, samples_per_pixel=1))
        with pytest.raises(abi.RtxError, match="upload_scene"):
            pt.render()
        pt.upload_scene(sd)
        with pytest.raises(abi.RtxError, match="set_camera"):
            pt.render()
        pt.set_camera(sd.camera(8, 8))
        pt.set_settings(sd.settings(width=8, height=8, samples_per_pixel=1, use_mnee=1, pipeline=abi.PIPELINE_WAVEFRONT))
        with pytest.raises(abi.RtxError, match="mnee"):
            pt.render()
        pt.set_settings(sd.settings(width=8, height=8, samples_per_pixel=0))
        with pytest.raises(abi.RtxError, match="bad settings"):
            pt.render()
        pt.set_settings(sd.settings(width=8, height=8, samples_per_pixel=1, shard_index=2, shard_count=2))
        with pytest.raises(abi.RtxError, match="shard"):
            pt.render()
        pt.set_settings(sd.settings(width=8, height=8, samples_per_pixel=1, mode=3))   # RenderMode::Spectral
        with pytest.raises(abi.RtxError, match="render mode 3"):
            pt.render()
        pt.set_settings(sd.settings(width=8, height=8, samples_per_pixel=1, mode=abi.MODE_BDPT, clamp_max=float("nan")))
        with pytest.raises(abi.RtxError, match="clamp_max"):
            pt.render()
        pt.set_settings(sd.settings(width=8, height=8, samples_per_pixel=2, max_depth=0, mode=abi.MODE_BDPT))   # no bounces: black, as in the reference
        assert float(np.abs(pt.render()).max()) == 0.0
        import copy
        nm = copy.copy(sd)
        nm._keep = []
        nm.materials = [copy.copy(m) for m in sd.materials]
        nm.materials[0].has_normal_map = 1
        nm.materials[0].normal_image = 5   # the scene has no images
        with pytest.raises(abi.RtxError, match="normal map index"):
            pt.upload_scene(nm)
        pt.upload_scene(sd)
        bad = copy.copy(sd)
        bad._keep = []
        s = abi.Sphere()
        s.center[:] = (0, 0, 0)
        s.radius = 1.0
        s.material_id = 99
        bad.spheres = list(sd.spheres) + [s]
        with pytest.raises(abi.RtxError, match="material id"):
            pt.upload_scene(bad)
        with pytest.raises(abi.RtxError, match="upload_scene"):   # a failed upload leaves no scene behind
            pt.render()
    with pytest.raises(abi.RtxError, match="out of range"):
        pkg.renderer.HipPathTracer(1 << 20)


@pytest.mark.parametrize("W,H,spp,shards", [(3, 2, 1, 1), (5, 4, 17, 1), (16, 9, 33, 1), (7, 5, 16, 2), (64, 33, 3, 3)])
@pytest.mark.parametrize("name", ["cornell_box_quad", "soup"])
def test_wavefront_pipeline_small_and_ragged_work(name, W, H, spp, shards, pkg, tracer, load_scene):
    """Films smaller than one workgroup of the queue pipeline, sample counts that are not a multiple of the chunk a slot sums
    (the last chunk of a pixel is shorter; a slot's last path of a work item sends its light sample to that item's sum), row
    shards: the queue pipeline still counts every sample once and agrees with the fused kernel."""
    sd = soup_scene(pkg, load_scene) if name == "soup" else load_scene(name)
    pt = tracer(sd)
    pt.set_camera(sd.camera(W, H))
    for shard in range(shards):
        rows = len(range(shard, H, shards))
        base = dict(width=W, height=H, samples_per_pixel=spp, max_depth=min(sd.max_depth, 12), seed=5, shard_index=shard, shard_count=shards)
        pt.set_settings(sd.settings(pipeline=pkg.abi.PIPELINE_FUSED, **base))
        a = pt.render().astype(np.float64)
        sa = pt.last_stats
        pt.set_settings(sd.settings(pipeline=pkg.abi.PIPELINE_WAVEFRONT, **base))
        b = pt.render().astype(np.float64)
        sb = pt.last_stats
        assert a.shape == b.shape == (rows, W, 3)
        assert sb.pipeline_used == pkg.abi.PIPELINE_WAVEFRONT and sb.samples == sa.samples == rows * W * spp
        assert abs(int(sb.segments) - int(sa.segments)) <= max(2, 0.01 * sa.segments) and abs(int(sb.rays_shadow) - int(sa.rays_shadow)) <= max(2, 0.01 * sa.rays_shadow)
        bad = (np.abs(a - b) > 1e-3 * (1.0 + np.abs(a))).any(axis=2)
        assert bad.sum() <= max(1, 0.01 * bad.size), (int(bad.sum()), float(np.abs(a - b).max()))


@pytest.mark.parametrize("force_mode", [None, "0", "1"])
@pytest.mark.parametrize("name", ["cornell_box", "cornell_box_quad", "environment_demo", "kitchen_sink", "textured", "soup", "normal_mapped", "sphere_field"])
def test_wavefront_pipeline_equals_fused(name, force_mode, pkg, tracer, load_scene, monkeypatch):
    """The queue-based pipeline runs the same per-path arithmetic in the same sample order as the
    fused kernel; only instruction selection (multiply-add fusion) may differ between the two
    compilations, so films agree to single-precision rounding and all counters are equal.
    RTX_FORCE_MODE (a diagnostic switch of the library) pushes every scene through the tree-walking
    kernel instances too (0: without, 1: with u/v resolution), whatever the scene would pick."""
    if force_mode is None:
        monkeypatch.delenv("RTX_FORCE_MODE", raising=False)
    else:
        monkeypatch.setenv("RTX_FORCE_MODE", force_mode)
    sd = soup_scene(pkg, load_scene) if name == "soup" else load_scene(name)
    W, H = 80, 45
    pt = tracer(sd)
    pt.set_camera(sd.camera(W, H))
    base = dict(width=W, height=H, samples_per_pixel=12, max_depth=min(sd.max_depth, 30), seed=77)
    pt.set_settings(sd.settings(pipeline=pkg.abi.PIPELINE_FUSED, **base))
    a = pt.render().astype(np.float64)
    sa = pt.last_stats
    pt.set_settings(sd.settings(pipeline=pkg.abi.PIPELINE_WAVEFRONT, **base))
    b = pt.render().astype(np.float64)
    sb = pt.last_stats
    assert sb.pipeline_used == pkg.abi.PIPELINE_WAVEFRONT and sa.pipeline_used == pkg.abi.PIPELINE_FUSED
    assert sb.samples == sa.samples == W * H * 12
    if name == "kitchen_sink":
        # its checker floor lies in the plane y = 0, where floor(y / scale) flips on the sign of a
        # rounding residue (see test_f64_film_equals_oracle): only means are comparable
        assert abs(a.mean() - b.mean()) < 0.015 * a.mean()
    else:
        # a rounding difference can flip a grazing intersection for a rare sample: allow a few pixels
        bad = (np.abs(a - b) > 1e-3 * (1.0 + np.abs(a))).any(axis=2)
        # (an environment map adds texel flips: a direction on a texel border picks the neighbour on a last-bit difference)
        assert bad.mean() < (0.01 if name == "environment_demo" else 0.004), (bad.mean(), np.abs(a - b).max())
        assert abs(a.mean() - b.mean()) < 2e-3 * (1 + a.mean())
    assert abs(int(sb.rays_closest) - int(sa.rays_closest)) <= 0.002 * sa.rays_closest
    assert abs(int(sb.rays_shadow) - int(sa.rays_shadow)) <= 0.002 * sa.rays_shadow
    # the wavefront pipeline is itself reproducible bit for bit
    c = pt.render().astype(np.float64)
    assert np.array_equal(b, c)


@pytest.mark.parametrize("mapper", [0, 1, 2, 3, 4])
def test_postprocess_bytes_equal_reference_and_oracle(mapper, pkg, oracle):
    """rtx_postprocess computes in double precision in the reference's operation order: its bytes equal the real
    reference's (fixture made by Image::apply_tone_mapping + Image::write_ppm) and the oracle's, exactly."""
    fx = np.load(os.path.join(GOLDEN, "post", "post.npz"))
    with pkg.renderer.HipPathTracer(0) as pt:
        for k, exposure in enumerate(fx["exposures"]):
            got = pt.postprocess(fx["film"], mapper, float(exposure))
            assert np.array_equal(got, fx[f"m{mapper}_e{k}"]), (mapper, exposure)
        # a 1080p film of rendered-like values against the oracle; NaN gives 0
        rng = np.random.default_rng(mapper)
        film = np.exp(rng.uniform(-10.0, 3.0, size=(1080, 1920, 3))).astype(np.float32)
        film[5, 7, 1] = np.nan
        got = pt.postprocess(film, mapper, 1.3)
        want = oracle.postprocess(film, mapper, 1.3)
        want[1080 - 1 - 5, 7, 1] = 0   # (int)NaN is undefined in the reference; defined as 0 here
        assert np.array_equal(got, want)


def test_postprocess_device_buffers_and_errors(pkg, oracle):
    import torch
    film = torch.rand(36, 64, 3, device="cuda", dtype=torch.float32) * 3.0
    out = torch.empty(36, 64, 3, device="cuda", dtype=torch.uint8)
    with pkg.renderer.HipPathTracer(0) as pt:
        torch.cuda.synchronize()
        pt.postprocess_device(film.data_ptr(), 64, 36, out.data_ptr(), pkg.scene_loader.TM_ACES, 0.8)
        assert np.array_equal(out.cpu().numpy(), oracle.postprocess(film.cpu().numpy(), pkg.scene_loader.TM_ACES, 0.8))
        with pytest.raises(pkg.abi.RtxError, match="tone mapper"):
            pt.postprocess(np.zeros((2, 2, 3), np.float32), 9, 1.0)


SIZES = ((1, 1, 1, 1), (1, 1, 5000, 3), (65535, 1, 1, 2), (1, 4097, 2, 0), (33, 17, 17, 1), (7, 5, 3, 100000))


def test_queue_pipeline_shard_of_a_very_large_film(pkg, tracer, load_scene):
    """The queue pipeline cuts a pixel's samples into up to 64 chunks, and a launch indexes fewer than 2^31 (pixel, chunk) work
    items: a film of more than 2^31 / 64 pixels takes fewer chunks (by the size of the whole film, so every shard of the frame
    agrees) instead of being refused.  One row of a 65535 x 600 film (shard 0 of 600), as two different launches of the same rows."""
    abi = pkg.abi
    sd = load_scene("kitchen_sink")
    pt = tracer(sd)
    W, H, spp = 65535, 600, 64
    pt.set_camera(sd.camera(W, H))
    kw = dict(width=W, height=H, samples_per_pixel=spp, max_depth=4, seed=3, pipeline=abi.PIPELINE_WAVEFRONT, precision=abi.PRECISION_F32)
    pt.set_settings(sd.settings(shard_index=0, shard_count=600, **kw))
    row = pt.render().copy()
    assert row.shape == (1, W, 3) and np.isfinite(row).all() and pt.last_stats.samples == W * spp
    pt.set_settings(sd.settings(shard_index=0, shard_count=300, **kw))   # rows 0 and 300
    two = pt.render()
    assert two.shape == (2, W, 3) and np.array_equal(two[0].view(np.uint32), row[0].view(np.uint32))


@pytest.mark.parametrize("name", ["cornell_box", "kitchen_sink", "normal_mapped", "environment_demo"])
def test_corner_sizes_all_modes(name, pkg, tracer, load_scene):
    """Corner sizes through every mode, pipeline and precision: 1x1 films, one row of the maximum width, one column taller
    than a workgroup, zero depth, sample counts far above the chunk size, depths far above any path's length.  Every render
    must come back finite and of the right shape (no hang, no fault, no NaN)."""
    abi = pkg.abi
    sd = load_scene(name)
    pt = tracer(sd)
    n = 0
    for (W, H, spp, depth) in SIZES:
        pt.set_camera(sd.camera(W, H))
        for mode in (abi.MODE_WHITTED, abi.MODE_PATHTRACE, abi.MODE_BDPT, abi.MODE_PATHPHOTON):
            for pipe in (abi.PIPELINE_FUSED, abi.PIPELINE_WAVEFRONT):
                for prec in (abi.PRECISION_F32, abi.PRECISION_F64):
                    if pipe == abi.PIPELINE_WAVEFRONT and (mode != abi.MODE_PATHTRACE or prec == abi.PRECISION_F64):
                        continue   # the queue pipeline is single-precision path tracing only
                    pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=spp, max_depth=depth, mode=mode, pipeline=pipe, precision=prec))
                    film = pt.render()
                    assert film.shape == (H, W, 3) and np.isfinite(film).all(), (W, H, spp, depth, mode, pipe, prec)
                    assert pt.last_stats.samples == W * H * spp
                    if depth == 0:
                        assert float(np.abs(film).max()) == 0.0 or mode == abi.MODE_WHITTED   # no bounce allowed: black
                    n += 1
    assert n == 6 * 9
