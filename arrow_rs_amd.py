"""Import shim: the package directory is ``arrow-rs_amd/`` (the name the project
layout prescribes), which Python cannot import by name.  ``import arrow_rs_amd``
loads that directory as a regular package under this module name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "arrow-rs_amd")
_spec = importlib.util.spec_from_file_location(
    "arrow_rs_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["arrow_rs_amd"] = _mod
_spec.loader.exec_module(_mod)
