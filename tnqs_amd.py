"""Import alias: the package directory is named `tensornetworkquantumsimulator.jl_amd` (with a dot, after the
reference repository), which Python cannot import by name -- load it from its path and expose it as `tnqs_amd`."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tensornetworkquantumsimulator.jl_amd")
_spec = importlib.util.spec_from_file_location("tnqs_amd", os.path.join(_pkg_dir, "__init__.py"),
                                               submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["tnqs_amd"] = _mod
_spec.loader.exec_module(_mod)
