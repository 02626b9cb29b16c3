"""Latency of the resident-fit entry points (bq_gp_fit / bq_gp_predict / alpha) on C2."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

e = Engine(0)
c = wl.c2()


def t(f, reps=20):
    f()
    e.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    e.sync()
    return (time.perf_counter() - t0) / reps * 1e3


fit = e.gp_fit(c["x"], c["y"], c["h"], c["w"], c["s"])
print("refit (gram+potrf+logml) %.3f ms" % t(lambda: fit.refit(c["h"], c["w"], c["s"])))
print("predict mean+var M=256   %.3f ms" % t(lambda: fit.predict(c["xo"])))
print("predict mean only        %.3f ms" % t(lambda: fit.predict(c["xo"], want_var=False)))
print("predict full cov         %.3f ms" % t(lambda: fit.predict(c["xo"], want_cov=True)))
print("alpha                    %.3f ms" % t(lambda: fit.alpha()))
print("fit_predict one-shot     %.3f ms" % t(lambda: e.fit_predict(c["x"], c["y"], c["h"], c["w"], c["s"], c["xo"])))
fit.close()
e.close()
