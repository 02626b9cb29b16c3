"""Many one-vector solves of many sizes through bq_gp_solve (kernel copies + one fill + one launch
per sweep) against numpy on the fit's own factor: residual check every call, wall time bounded."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

e = Engine(0)
rs = np.random.RandomState(1)
t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else time.time() + 40.0
fits = {}
for n in (1000, 1024, 1100, 1536, 2048, 2500, 4096, 5000):
    c = wl.c4(n)
    fits[n] = (e.gp_fit(c["x"], wl.norm_logpdf(c["x"]), c["h"], c["w"] * 3.0, c["s"]), None)
calls = 0
worst = 0.0
while time.time() < t_end:
    n = int(rs.choice(list(fits)))
    fit, K = fits[n]
    if K is None:
        K = fit.K()
        fits[n] = (fit, K)
    b = rs.randn(n)
    x = fit.solve(b)
    r = np.max(np.abs(K.dot(x) - b)) / (np.max(np.abs(K)) * np.max(np.abs(x)) * n)
    worst = max(worst, r)
    assert np.isfinite(x).all() and r < 1e-13, (n, r)
    calls += 1
    if calls % 500 == 0:
        print("calls", calls, "worst scaled residual %.1e" % worst, flush=True)
print("done: calls", calls, "worst scaled residual %.1e" % worst, flush=True)
