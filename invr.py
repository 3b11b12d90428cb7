"""Import shim: ``import invr`` loads the package that lives in ``instant-nvr_amd/``.

The directory name required by the repo layout contains a hyphen and is therefore
not a Python identifier; this file registers that directory as the package
``invr`` (its ``__name__`` is ``invr``, sub-modules are ``invr.<x>``).
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "instant-nvr_amd")
_spec = importlib.util.spec_from_file_location(
    "invr", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["invr"] = _mod
_spec.loader.exec_module(_mod)
