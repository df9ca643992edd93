"""Import shim: `import mtl_amd` loads the package in `meta-transfer-learning_amd/` (a hyphen is not importable)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'meta-transfer-learning_amd')
_spec = importlib.util.spec_from_file_location('mtl_amd', os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['mtl_amd'] = _mod
_spec.loader.exec_module(_mod)
