"""la.cho_solve_mat (square B, linalg_c.pyx:139-179) through the drop-in module: wall time per
call at several n, and the device kernels' share for the resident-factor entry point."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import la  # noqa: E402

for n in (64, 512, 2048, 4096):
    rs = np.random.RandomState(n)
    G = rs.randn(n, 16)
    A = np.asfortranarray(G.dot(G.T) / 16 + np.eye(n))
    L = np.asfortranarray(np.linalg.cholesky(A))
    B = np.asfortranarray(rs.randn(n, n))
    X = np.empty_like(B, order="F")
    la.cho_solve_mat(L, B, X)
    t0 = time.perf_counter()
    for _ in range(3):
        la.cho_solve_mat(L, B, X)
    dt = (time.perf_counter() - t0) / 3
    res = np.abs(A.dot(X) - B).max()
    print("n=%d cho_solve_mat %.2f ms  (%.1f GFLOP/s on 2 n^3)  residual %.1e"
          % (n, dt * 1e3, 2.0 * n ** 3 / dt / 1e9, res), flush=True)
