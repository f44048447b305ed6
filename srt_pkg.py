"""Registers ./simple-raytracer_amd (hyphenated directory) as package `simple_raytracer_amd`."""
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
_NAME = "simple_raytracer_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    pkg_dir = ROOT / "simple-raytracer_amd"
    spec = importlib.util.spec_from_file_location(_NAME, pkg_dir / "__init__.py", submodule_search_locations=[str(pkg_dir)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
