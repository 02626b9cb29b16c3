"""The hyper-parameter loop's body on resident fits (bench.py hyper_loop_body) with the small
transfers through kernels on the mapped staging buffer (BQ_SOLVE_KCOPY=1, default) or through the
copy engine: ms per iteration and the values."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

for kc in (0, 1):
    os.environ["BQ_SOLVE_KCOPY"] = str(kc)
    eng = Engine(0)
    for n, nc in ((19, 10), (100, 10), (1024, 10)):
        dx = 10.0 / (n - 1)
        x = np.linspace(-5.0, 5.0, n)
        xc = np.linspace(-5.5, 5.5, nc) + 0.37 * dx
        g1 = eng.gp_fit(x, wl.norm_logpdf(x), 15.0, 1.3 * dx, 1e-3)
        xsc = np.concatenate([x, xc])
        g2 = eng.gp_fit(xsc, np.exp(wl.norm_logpdf(xsc)), 0.2, 1.3 * dx, 1e-3)
        ts = []
        acc = 0.0
        for rep in range(5):
            t0 = time.perf_counter()
            for it in range(50):
                wi = (1.3 + 0.001 * it) * dx
                m, v = g1.refit_predict(15.0, wi, 1e-3, xc)
                g2.refit(0.2, wi, 1e-3)
                acc = g1.logml + g2.logml + m.sum() + v.sum()
            ts.append((time.perf_counter() - t0) / 50 * 1e3)
        print("kcopy %d n %4d: %.4f ms per iteration, last value %.15e" % (kc, n, sorted(ts)[2], acc),
              flush=True)
        g1.close()
        g2.close()
    eng.close()
