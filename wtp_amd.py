"""Import shim: the package directory is literally `whatsthepoint.jl_amd/` (a dot in the
name), which `import` cannot spell, so it is loaded here under the module name
`whatsthepoint_jl_amd` and re-exported as `wtp_amd`."""
import importlib.util
import os
import sys

_NAME = "whatsthepoint_jl_amd"
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "whatsthepoint.jl_amd")

if _NAME not in sys.modules:
    _spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR]
    )
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[_NAME] = _mod
    _spec.loader.exec_module(_mod)

_pkg = sys.modules[_NAME]
globals().update({k: v for k, v in vars(_pkg).items() if not k.startswith("__")})
__all__ = getattr(_pkg, "__all__", [])
