"""Single right-hand-side solve on resident factors: wall time per call and (under rocprofv3)
the per-kernel durations of the GEMV sweeps."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from bayesian_quadrature_amd import Engine, workloads as wl  # noqa: E402

e = Engine(0)
for n in (4096, 16384):
    c = wl.c4(n)
    y = wl.norm_logpdf(c["x"])
    fit = e.gp_fit(c["x"], y, c["h"], c["w"], c["s"])
    b = np.random.RandomState(n).randn(n)
    fit.solve(b)
    e.sync()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(20):
            fit.solve(b)
        e.sync()
        ts.append((time.perf_counter() - t0) / 20 * 1e3)
    print(n, "ms per solve (wall, host buffers): min %.4f median %.4f max %.4f"
          % (min(ts), sorted(ts)[3], max(ts)), flush=True)
    fit.close()
e.close()
