"""Import shim: the package directory is named ``adapt-image-models_amd`` (not a valid Python
identifier), so ``import aim_amd`` loads it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "adapt-image-models_amd")
_spec = importlib.util.spec_from_file_location(
    "aim_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["aim_amd"] = _mod
_spec.loader.exec_module(_mod)
