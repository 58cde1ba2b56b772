"""Importable alias of the package directory ``osvos-pytorch_amd/`` (a hyphen is not a valid
Python identifier, so this shim points the package search path at it and runs its __init__)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "osvos-pytorch_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
