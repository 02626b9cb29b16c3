"""The one-launch single-vector sweeps (trsvflow.h) against the one-launch-per-block ones
(BQ_TRSV_FLOW=0): same bits, and the wall time of fit.solve(b) for one right-hand side."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402


def run(flow, sizes):
    os.environ["BQ_TRSV_FLOW"] = "1" if flow else "0"
    e = Engine(0)
    del os.environ["BQ_TRSV_FLOW"]
    out = {}
    for n in sizes:
        c = wl.c4(n)
        y = wl.norm_logpdf(c["x"])
        fit = e.gp_fit(c["x"], y, c["h"], c["w"] * 3.0, c["s"])
        rs = np.random.RandomState(n)
        b = rs.randn(n)
        x = fit.solve(b)
        a = fit.alpha()
        for _ in range(3):
            fit.solve(b)
        e.sync()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(10):
                fit.solve(b)
            ts.append((time.perf_counter() - t0) / 10 * 1e3)
        out[n] = (x, a, sorted(ts)[2])
        fit.close()
    e.close()
    return out


sizes = [int(v) for v in sys.argv[1:]] or [512, 1000, 1536, 2048, 2500, 4096, 6000, 16384]
ref = run(False, sizes)
new = run(True, sizes)
for n in sizes:
    same = np.array_equal(ref[n][0], new[n][0]) and np.array_equal(ref[n][1], new[n][1])
    print("n %5d: identical %s  finite %s  per-block launches %.3f ms  one launch per sweep %.3f ms"
          % (n, same, bool(np.isfinite(new[n][0]).all()), ref[n][2], new[n][2]), flush=True)
