"""The hyper-parameter loop body on a resident fit: refit, then the posterior of
a few candidate points (mean + variance), repeated -- every predict pays for the factor's block
inverses again."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine, workloads as wl  # noqa: E402

e = Engine(0)
for n in (19, 256, 1024):
    x = np.linspace(-5, 5, n)
    y = wl.norm_logpdf(x)
    dx = 10.0 / (n - 1)
    fit = e.gp_fit(x, y, 1.0, 1.3 * dx, 1e-3)
    xo = np.linspace(-6, 6, 10)
    for phase in ("refit only", "refit + predict(10)", "refit_predict(10)", "refit + alpha"):
        ts = []
        for rep in range(5):
            e.sync()
            t0 = time.perf_counter()
            for it in range(20):
                if phase.startswith("refit_predict"):
                    fit.refit_predict(1.0, (1.3 + 0.001 * it) * dx, 1e-3, xo)
                    continue
                fit.refit(1.0, (1.3 + 0.001 * it) * dx, 1e-3)
                if phase.endswith("(10)"):
                    fit.predict(xo)
                elif phase.endswith("alpha"):
                    fit.alpha()
            e.sync()
            ts.append((time.perf_counter() - t0) / 20 * 1e3)
        print("n=%d %-22s %.3f ms" % (n, phase, sorted(ts)[2]), flush=True)
    fit.close()
e.close()
