"""Wall time of fit.solve(b) with ONE right-hand side on resident factors (hipGraph replay of
the GEMV sweeps, host vector in and out)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

e = Engine(0)
for n in (1024, 4096, 16384):
    c = wl.c4(n)
    fit = e.gp_fit(c["x"], wl.norm_logpdf(c["x"]), c["h"], c["w"], c["s"])
    b = np.random.RandomState(n).randn(n)
    fit.solve(b)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(20):
            fit.solve(b)
        ts.append((time.perf_counter() - t0) / 20 * 1e3)
    print("n=%d: %.4f ms per solve -> %.0f GB/s on 8 N^2 bytes" % (n, sorted(ts)[3], 8.0 * n * n / sorted(ts)[3] / 1e6))
    fit.close()
e.close()
