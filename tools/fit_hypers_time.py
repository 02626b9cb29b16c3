"""One evaluation of the hyper-parameter objective of BQ.fit_hypers (bq.py:536-550: set GP1's
parameters, re-predict the candidates, set GP2's targets and parameters, both log-MLs) on the
reference's own fixture size and on larger sample sets, through the device engine."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesian_quadrature_amd as bqa  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

for n in (9, 64, 512):
    x = np.linspace(-5, 5, n)
    l = np.exp(wl.norm_logpdf(x))
    b = bqa.BQ(x, l, n_candidate=10, x_mean=0.0, x_var=10.0, candidate_thresh=0.5,
               kernel=bqa.GaussianKernel, optim_method="L-BFGS-B")
    np.random.seed(8728)
    dx = 10.0 / (n - 1)
    if n == 9:
        b.init(params_tl=(15.0, 2.0, 0.0), params_l=(0.2, 1.3, 0.0))
    else:
        b.init(params_tl=(15.0, 1.3 * dx, 1e-3), params_l=(0.2, 1.3 * dx, 1e-4))
    f = b._make_llh_params(["h", "w"])
    p0 = b._current_params(["h", "w"])
    f(p0)
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        for it in range(20):
            f(p0 * (1.0 + 1e-4 * (it + 1)))
        ts.append((time.perf_counter() - t0) / 20 * 1e3)
    print("n=%d (nc=%d): %.3f ms per objective evaluation" % (n, b.nc, sorted(ts)[2]), flush=True)
