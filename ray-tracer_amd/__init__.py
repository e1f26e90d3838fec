"""MI355X-native path-tracing backend: host-side mirror of the reference interface.

The directory name carries a hyphen, so import it through ``rtx_pkg.load()`` (repo root),
which registers it as the module ``raytracer_amd``.
"""
from . import abi, scene_loader, renderer, synthetic  # noqa: F401  (parallel imports torch: import it explicitly)
