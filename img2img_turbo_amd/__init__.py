"""Importable alias of the ``img2img-turbo_amd/`` package directory.

The package directory carries the reference repo's name (with its hyphen); Python cannot import a
hyphenated name, so this stub points its ``__path__`` there and runs the real ``__init__``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "img2img-turbo_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
