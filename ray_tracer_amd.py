"""Import alias: makes the package directory `ray-tracer_amd/` (not a valid Python identifier) importable as
`ray_tracer_amd`."""
import importlib.util
import sys
from pathlib import Path

_pkg_dir = Path(__file__).resolve().parent / "ray-tracer_amd"
_spec = importlib.util.spec_from_file_location("ray_tracer_amd", _pkg_dir / "__init__.py",
                                               submodule_search_locations=[str(_pkg_dir)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ray_tracer_amd"] = _mod
_spec.loader.exec_module(_mod)
