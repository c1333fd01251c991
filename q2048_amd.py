"""Import shim: `import q2048_amd` == the package in ./2048_q-learning_amd/ (whose directory
name BASELINE fixes and which is not a valid Python identifier)."""
import importlib as _importlib
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.abspath(__file__))
if _here not in _sys.path:
    _sys.path.insert(0, _here)
_pkg = _importlib.import_module("2048_q-learning_amd")
globals().update({k: getattr(_pkg, k) for k in _pkg.__all__})
package = _pkg
__all__ = list(_pkg.__all__) + ["package"]
