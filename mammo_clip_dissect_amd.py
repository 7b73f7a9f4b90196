"""Import alias: `import mammo_clip_dissect_amd` loads the package in ./mammo-clip-dissect_amd/.

The package directory carries the project's name (with hyphens), which is not a Python
identifier; this one-file shim registers it under an importable name.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mammo-clip-dissect_amd")
_spec = importlib.util.spec_from_file_location(
    "mammo_clip_dissect_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mammo_clip_dissect_amd"] = _mod
_spec.loader.exec_module(_mod)
