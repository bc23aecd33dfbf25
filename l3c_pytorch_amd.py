"""Import alias: `import l3c_pytorch_amd` loads the package that lives in `l3c-pytorch_amd/`.

The package directory carries the project's hyphenated name, which is not a Python identifier; this one-file
module registers it under an importable name (and is replaced in sys.modules by the real package object).
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'l3c-pytorch_amd')
_spec = importlib.util.spec_from_file_location(
    'l3c_pytorch_amd', os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['l3c_pytorch_amd'] = _mod
_spec.loader.exec_module(_mod)
