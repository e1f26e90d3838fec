import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import rtx_pkg
    return rtx_pkg.load()


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def load_scene(pkg):
    cache = {}

    def _load(name):
        if name not in cache:
            cache[name] = pkg.scene_loader.load(os.path.join(GOLDEN, "scenes", name + ".json"))
        return cache[name]
    return _load


C5_TRIANGLES = 10_000_000


@pytest.fixture(scope="session")
def c5(pkg, load_scene):
    """The synthetic BVH stress configuration at full size (10 M triangles): stress_base.json (the enclosure the reference's own
    loader read when hits/c5.npz was made) + the seeded triangle soup, with the fixture's two ray sets regenerated from their
    seeds."""
    import copy
    fx = np.load(os.path.join(GOLDEN, "hits", "c5.npz"))
    assert int(fx["n_triangles"]) == C5_TRIANGLES
    sd = copy.copy(load_scene("stress_base"))
    sd._keep = []
    sd.triangle_array = pkg.synthetic.triangle_soup(C5_TRIANGLES, seed=int(fx["soup_seed"]))
    W, H = int(fx["width"]), int(fx["height"])
    cam = sd.camera(W, H)
    got_cam = np.array([cam.origin[:], cam.lower_left_corner[:], cam.horizontal[:], cam.vertical[:]])
    assert np.array_equal(got_cam, fx["camera"]), "camera vectors differ from the ones the fixture's pixel rays were made with"
    rays = {"random": pkg.synthetic.stress_rays(int(fx["n_random"]), seed=int(fx["ray_seed"])),
            "pixels": pkg.synthetic.pixel_centre_rays(cam, W, H)}
    return {"scene": sd, "fx": fx, "rays": rays}


def record_rate(name, **values):
    """Append one measured figure (a mismatch rate, a relMSE ...) to gpurun_out/parity_rates.jsonl, from where the numbers the
    thresholds are set against are copied into profiles/."""
    import json
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_rates.jsonl"), "a") as f:
            f.write(json.dumps(dict(name=name, **values)) + "\n")
    except OSError:
        pass


def soup_scene(pkg, load_scene_fn, base="cornell_box_quad", n=4000):
    """cornell_box_quad + the 4000-triangle soup the hits/soup.npz fixture was made with (or another base / size)."""
    import copy
    base = load_scene_fn(base)
    sd = copy.copy(base)
    sd._keep = []
    soup = pkg.synthetic.triangle_soup(n, seed=7, extent=2.2, size=0.25, n_materials=5)
    for k in ("v0", "v1", "v2"):
        soup[k] += (0, 2.5, -2.5)
    sd.triangle_array = soup
    return sd


def rel_mse(x, r):
    """relMSE on the clamped-linear film (Renderer::render never returns unclamped radiance,
    engine/renderer.cpp:175-182)."""
    x = np.clip(np.asarray(x, dtype=np.float64), 0.0, 1.0)
    r = np.clip(np.asarray(r, dtype=np.float64), 0.0, 1.0)
    return float(np.mean((x - r) ** 2 / (r ** 2 + 1e-2)))
