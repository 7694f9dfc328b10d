"""Import shim: `import odtk` -> the package in ./object-detection-tensorflow_amd/
(a hyphenated directory name cannot be imported directly)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "object-detection-tensorflow_amd")
_spec = importlib.util.spec_from_file_location("odtk", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["odtk"] = _mod
_spec.loader.exec_module(_mod)
