"""Import alias: the package directory is named `ctc-vr_amd` (not a Python identifier).

`import ctc_vr_amd` executes this file, which loads `ctc-vr_amd/__init__.py` as the package
`ctc_vr_amd` (with its directory as the submodule search path) and replaces itself in
sys.modules, so `import ctc_vr_amd.lib` etc. work as usual.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ctc-vr_amd")
_spec = importlib.util.spec_from_file_location(
    "ctc_vr_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ctc_vr_amd"] = _mod
_spec.loader.exec_module(_mod)
