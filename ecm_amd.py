"""Import shim: the package directory `explicit-context-mapping-for-stereo-matching_amd/` is not a valid
identifier, so load it with importlib and re-export it as `ecm_amd`."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_pkg = importlib.import_module("explicit-context-mapping-for-stereo-matching_amd")
globals().update({k: getattr(_pkg, k) for k in _pkg.__all__})
_lib = _pkg._lib
models = importlib.import_module("explicit-context-mapping-for-stereo-matching_amd.models")
__all__ = list(_pkg.__all__) + ["models"]
