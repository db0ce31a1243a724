"""Import shim: makes the in-tree directory ``recursivefactorization.jl_amd/`` importable as the Python package
``recursivefactorization.jl_amd`` (a directory name with a dot cannot be found by the default importer)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "recursivefactorization.jl_amd")
_name = __name__ + ".jl_amd"
if _name not in _sys.modules:
    _spec = _ilu.spec_from_file_location(_name, _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
    _mod = _ilu.module_from_spec(_spec)
    _sys.modules[_name] = _mod
    _spec.loader.exec_module(_mod)
jl_amd = _sys.modules[_name]
