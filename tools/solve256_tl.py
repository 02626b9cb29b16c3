"""Per-launch durations (HIP events, eager) of fit.solve(B) with 256 right-hand sides at size n."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
e = Engine(0)
c = wl.c4(n)
fit = e.gp_fit(c["x"], wl.norm_logpdf(c["x"]), c["h"], c["w"], c["s"])
B = np.asfortranarray(np.random.RandomState(0).randn(n, 256))
for _ in range(3):
    fit.solve(B)
rows = e.timeline(lambda: fit.solve(B))
print("launches", len(rows), "durations us:", " ".join("%.1f" % ((r[3] - r[2]) * 1e3) for r in rows))
fit.close()
e.close()
