"""Importable alias of the ``gfx-ocean_amd/`` package directory (a hyphen cannot be imported).

``import gfx_ocean_amd`` executes ``gfx-ocean_amd/__init__.py`` with this module's
``__path__`` pointing at that directory, so ``gfx_ocean_amd.ocean`` is
``gfx-ocean_amd/ocean.py`` and so on.  No code lives here.
"""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gfx-ocean_amd")
__path__ = [_REAL]
with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
del _f
