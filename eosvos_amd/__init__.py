"""Importable alias for the `e-osvos_amd/` package directory.

The product package lives in `e-osvos_amd/` (the name the repository layout
prescribes); a hyphen is not a valid Python identifier, so this stub makes the same
directory importable as `eosvos_amd`: sub-modules (`eosvos_amd.engine`, ...) resolve
to `e-osvos_amd/*.py` through `__path__`.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'e-osvos_amd')
__path__ = [_real]
with open(_os.path.join(_real, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))
del _f
