"""MI355X-native Bayesian-quadrature GP engine.

Python host code over hand-written HIP (gfx950) behind the C ABI of
``include/bqhip.h``.  The package mirrors the reference's import surface for
its hot path: ``BQ`` (bayesian_quadrature/bq.py), ``linalg`` (linalg_c.pyx) and
the ``gp`` objects the reference imports from the third-party ``gp`` package.

Importing the package does not touch the GPU; the first operation does, and it
raises if ``libbqhip.so`` or a HIP device is missing (no CPU fallback).
"""
from . import _lib  # noqa: F401
from . import engine  # noqa: F401
from . import linalg  # noqa: F401
from . import linalg as la  # noqa: F401  (the reference's alias, bq.py:10)
from .engine import Engine, get_engine, set_engine  # noqa: F401
from .pool import EnginePool  # noqa: F401
from .gp import GP, GaussianKernel, PeriodicKernel  # noqa: F401
from . import gauss, bq_c, util  # noqa: F401
from . import gauss as gauss_c  # noqa: F401  (the reference's module name)
from .bq import BQ  # noqa: F401

__version__ = "0.1.0"
