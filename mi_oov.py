"""Importable alias of the package directory `improving-inductive-oov-recsys_amd/` (whose name is
not a valid Python identifier): `import mi_oov` returns that package."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("improving-inductive-oov-recsys_amd")
sys.modules[__name__] = _pkg
