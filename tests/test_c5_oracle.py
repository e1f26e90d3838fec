"""The 10 M-triangle stress configuration on the CPU: the oracle's closest-hit tables equal the real reference's
(hits/c5.npz, made by oracle/_ref/ref_harness: digest of every record + the first 10 000 verbatim)."""
import numpy as np

from conftest import C5_TRIANGLES


def check_against_fixture(pkg, fx, name, hits):
    first = fx[name + "_first"]
    for f in ("hit", "prim_type", "prim_index", "material_id", "front_face"):
        assert np.array_equal(hits[f][: len(first)], first[f]), (name, f)
    for f in ("t", "point", "normal"):
        assert np.array_equal(np.ascontiguousarray(hits[f][: len(first)]).view(np.uint64), np.ascontiguousarray(first[f]).view(np.uint64)), (name, f)
    assert int(hits["hit"].sum()) == int(fx[name + "_n_hit"]) and int((hits["prim_type"] == 2).sum()) == int(fx[name + "_n_tri"])
    assert pkg.synthetic.hit_table_digest(hits) == str(fx[name + "_digest"]), name


def test_oracle_equals_reference_on_the_stress_scene(pkg, oracle, c5):
    osc = oracle.OracleScene(c5["scene"])
    boxes, nodes, prims = osc.bvh()
    assert len(prims) == C5_TRIANGLES and len(nodes) > C5_TRIANGLES
    del boxes, nodes, prims
    for name, rays in c5["rays"].items():
        check_against_fixture(pkg, c5["fx"], name, osc.trace_closest(rays, threads=16))
    osc.close()
