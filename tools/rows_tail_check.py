"""A large row sweep's last updates as split-k tiles (BQ_ROWS_TAIL=<LDS tiles>): the 256-RHS solve at
size n against the default, values and per-launch durations."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
out = {}
for tail in [int(v) for v in sys.argv[2:]] or [0, 64, 128, 192]:
    os.environ["BQ_ROWS_TAIL"] = str(tail)
    e = Engine(0)
    c = wl.c4(n)
    fit = e.gp_fit(c["x"], wl.norm_logpdf(c["x"]), c["h"], c["w"], c["s"])
    B = np.asfortranarray(np.random.RandomState(0).randn(n, 256))
    for _ in range(3):
        X = fit.solve(B)
    rows = e.timeline(lambda: fit.solve(B))
    d = [(r[3] - r[2]) * 1e3 for r in rows]
    out[tail] = X
    print("tail %3d: %d launches, kernels %.1f us: %s   max |X - X0| / max |X0| %.1e" % (
        tail, len(rows), sum(d), " ".join("%.0f" % v for v in d),
        np.max(np.abs(X - out[0])) / np.max(np.abs(out[0])) if 0 in out else 0.0), flush=True)
    fit.close()
    e.close()
