"""Import shim: the package directory is ``cm-tts_amd/`` (a hyphen cannot appear in a Python
module name), so ``import cmtts_amd`` loads that directory as the package ``cmtts_amd``."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "cm-tts_amd")
_spec = importlib.util.spec_from_file_location(
    "cmtts_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["cmtts_amd"] = _mod
_spec.loader.exec_module(_mod)
