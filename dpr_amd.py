"""Import alias: the package directory is named `diffpointrasterisation.jl_amd` (with a
dot, after the reference), which Python's import statement cannot spell.  `import dpr_amd`
loads that directory as the module `diffpointrasterisation_jl_amd` and re-exports it."""
import importlib.util
import os
import sys

_NAME = "diffpointrasterisation_jl_amd"
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diffpointrasterisation.jl_amd")


def _load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod


_pkg = _load()
globals().update({k: getattr(_pkg, k) for k in _pkg.__all__})
_lib = _pkg._lib
interface = sys.modules[_NAME + ".interface"]
sharded = sys.modules[_NAME + ".sharded"]
__all__ = list(_pkg.__all__)
