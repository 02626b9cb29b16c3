"""What the width of the explicit diagonal-block inverses (wide_block, BQ_WIDE_B) costs and buys on
resident fits of 1024 / 1536 points: refit + predict (the inverses are rebuilt after every refit),
predict alone, one-vector solve, 256-vector solve.  python tools/wide_b_check.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

e = Engine(0)


def t(f, reps=30):
    f()
    e.sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        e.sync()
        ts.append((time.perf_counter() - t0) / reps * 1e3)
    return sorted(ts)[2]


for n in (1024, 1536):
    c = wl.c2(n=n)
    fit = e.gp_fit(c["x"], c["y"], c["h"], c["w"], c["s"])
    b = np.random.RandomState(0).randn(n)
    B = np.asfortranarray(np.random.RandomState(1).randn(n, 256))

    def refit_predict():
        fit.refit(c["h"], c["w"], c["s"])
        fit.predict(c["xo"])

    print("n=%d refit+predict %.3f  predict %.3f  solve1 %.3f  solve256 %.3f  refit+solve1 %.3f ms"
          % (n, t(refit_predict, 10), t(lambda: fit.predict(c["xo"])), t(lambda: fit.solve(b)),
             t(lambda: fit.solve(B), 5), t(lambda: (fit.refit(c["h"], c["w"], c["s"]), fit.solve(b)), 10)))
    fit.close()
e.close()
