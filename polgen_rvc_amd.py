"""Import shim: the package directory is ``polgen-rvc_amd/`` (a hyphen is not a
valid Python identifier), so ``import polgen_rvc_amd`` loads it from that
directory and replaces this module with the real package."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "polgen-rvc_amd")
_spec = importlib.util.spec_from_file_location(
    "polgen_rvc_amd",
    os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir],
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["polgen_rvc_amd"] = _mod
_spec.loader.exec_module(_mod)
