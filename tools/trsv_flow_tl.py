"""Per-launch timeline (HIP events, eager launches) of one fit.solve(b) at size n, both sweep forms."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for flow in (0, 1):
    os.environ["BQ_TRSV_FLOW"] = str(flow)
    e = Engine(0)
    c = wl.c4(n)
    fit = e.gp_fit(c["x"], wl.norm_logpdf(c["x"]), c["h"], c["w"] * 3.0, c["s"])
    b = np.random.RandomState(0).randn(n)
    for _ in range(3):
        fit.solve(b)
    rows = e.timeline(lambda: fit.solve(b))
    print("flow", flow, "launches", len(rows), "span %.1f us" % (max(r[3] for r in rows) * 1e3),
          "durations us:", " ".join("%.1f" % ((r[3] - r[2]) * 1e3) for r in rows))
    fit.close()
    e.close()
