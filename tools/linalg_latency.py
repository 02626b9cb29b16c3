"""Wall time per call of the linalg_c drop-ins on host buffers (cho_factor, cho_solve_vec, cho_solve_mat,
logdet) at the reference's own matrix sizes and a few larger ones."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import linalg as la  # noqa: E402


def t(fn, reps=200):
    fn()
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


rs = np.random.RandomState(0)
for n in (5, 30, 100, 513, 1024):
    A = rs.rand(n, n)
    C = np.asfortranarray(A + A.T + n * np.eye(n))
    L = np.empty_like(C)
    b = rs.randn(n)
    B = np.asfortranarray(rs.randn(n, n))
    x = np.empty(n)
    X = np.empty_like(B)
    la.cho_factor(C, L)
    reps = 200 if n <= 513 else 50
    print("n %4d: cho_factor %.3f ms  cho_solve_vec %.3f ms  cho_solve_mat(n x n) %.3f ms  logdet %.3f ms" % (
        n, t(lambda: la.cho_factor(C, L), reps), t(lambda: la.cho_solve_vec(L, b, x), reps),
        t(lambda: la.cho_solve_mat(L, B, X), max(reps // 4, 10)), t(lambda: la.logdet(L), reps)),
        flush=True)
