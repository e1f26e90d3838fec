"""Import helper: registers the package directory ``ray-tracer_amd/`` (not a valid Python
identifier) under the module name ``raytracer_amd``."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    if "raytracer_amd" in sys.modules:
        return sys.modules["raytracer_amd"]
    pkg_dir = os.path.join(_ROOT, "ray-tracer_amd")
    spec = importlib.util.spec_from_file_location(
        "raytracer_amd", os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["raytracer_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
