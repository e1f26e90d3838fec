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


@pytest.fixture(scope="session", autouse=True)
def _torch_device_first(request):
    """On a GPU box, let torch create its HIP context before the library's first kernel: a few tests hand torch
    tensors to the library, and torch's lazy initialisation was seen to fail ("No HIP GPUs are available") when it
    came after a long series of the library's own launches in the same process."""
    markexpr = request.config.getoption("-m", default="") or ""
    if "not gpu" in markexpr:
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda")
    except Exception:
        pass


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
