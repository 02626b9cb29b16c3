"""Wall time of the BQ-object operations on the reference's own fixture size (N = 9 + candidates)
and on a larger 1-D problem, through the device engine."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesian_quadrature_amd as bqa  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402


def make(n):
    x = np.linspace(-5, 5, n)
    l = np.exp(wl.norm_logpdf(x))
    opts = dict(n_candidate=10, x_mean=0.0, x_var=10.0, candidate_thresh=0.5,
                kernel=bqa.GaussianKernel, optim_method="L-BFGS-B")
    b = bqa.BQ(x, l, **opts)
    np.random.seed(8728)
    dx = 10.0 / (n - 1)
    if n == 9:   # the reference's own fixture (tests/util.py:43)
        b.init(params_tl=(15.0, 2.0, 0.0), params_l=(0.2, 1.3, 0.0))
    else:
        b.init(params_tl=(15.0, 1.3 * dx, 1e-3), params_l=(0.2, 1.3 * dx, 1e-4))
    return b


for n in (9, 64, 256):
    t0 = time.perf_counter()
    b = make(n)
    t1 = time.perf_counter()
    zm = b.Z_mean()
    t2 = time.perf_counter()
    zv = b.Z_var()
    t3 = time.perf_counter()
    xa = np.linspace(-6, 6, 50)
    esm = b.expected_squared_mean(xa)
    t4 = time.perf_counter()
    m = b.l_mean(xa)
    v = b.l_var(xa)
    t5 = time.perf_counter()
    print("n=%d init %.1f ms | Z_mean %.2f ms | Z_var %.2f ms | esm(50) %.1f ms | l_mean+l_var(50) %.2f ms | Z=%.6g V=%.3g"
          % (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, zm, zv))
