"""Import alias: the package directory is `nerfstudio-thermal_amd/` (not a valid Python identifier), so
`import nerfstudio_thermal_amd` loads it from there and registers it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nerfstudio-thermal_amd")
_spec = importlib.util.spec_from_file_location(
    "nerfstudio_thermal_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["nerfstudio_thermal_amd"] = _mod
_spec.loader.exec_module(_mod)
