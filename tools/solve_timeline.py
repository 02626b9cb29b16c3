"""Launch timeline of fit.solve with 256 right-hand sides on a resident N = 16384 (or argv[1])
factor: per launch class, start, duration (the engine's launch profiler)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
e = Engine(0)
c = wl.c4(n)
fit = e.gp_fit(c["x"], wl.norm_logpdf(c["x"]), c["h"], c["w"], c["s"])
B = np.asfortranarray(np.random.RandomState(1).randn(n, 256))
fit.solve(B)
rows = e.timeline(lambda: fit.solve(B))
print("launches", len(rows), "span %.3f ms" % max(r[3] for r in rows))
for cls, st, t0, t1, w in rows:
    print("%d %-14s %8.3f %8.1f us  %6.1f TFLOP/s" % (st, cls, t0, (t1 - t0) * 1e3,
                                                    w / max(t1 - t0, 1e-9) / 1e9))
fit.close()
e.close()
