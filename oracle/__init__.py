"""CPU parity oracle (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product package
(``bayesian_quadrature_amd``) never does.
"""
from .oracle import Oracle, load, build  # noqa: F401
