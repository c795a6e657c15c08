"""Import helper: the package directory is named ``climatemachine.jl_amd`` (a dot
is not a legal module name), so it is registered under the alias
``climatemachine_jl_amd``.  Usage: ``from cmdg_loader import cm``."""
import importlib.util
import os
import sys

_ALIAS = "climatemachine_jl_amd"
_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.join(_ROOT, "climatemachine.jl_amd")


def load():
    if _ALIAS in sys.modules:
        return sys.modules[_ALIAS]
    spec = importlib.util.spec_from_file_location(
        _ALIAS, os.path.join(_PKG, "__init__.py"),
        submodule_search_locations=[_PKG])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod


cm = load()
