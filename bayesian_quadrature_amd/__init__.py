"""Importable alias of the ``bayesian-quadrature_amd/`` source directory.

The product package lives in ``bayesian-quadrature_amd/`` (a name Python cannot
import because of the hyphen); this stub points the package search path there
and executes its ``__init__``.
"""
import os as _os

_src = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                     "bayesian-quadrature_amd")
__path__ = [_src]
with open(_os.path.join(_src, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_src, "__init__.py"), "exec"))
del _os, _f
