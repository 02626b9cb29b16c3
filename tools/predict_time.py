"""bq_gp_predict mean + variance on a resident N = 1024 fit: wall per call and the kernels'
HIP-event time per class (bench.py's _prof_call): python tools/predict_time.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

e = Engine(0)
c2 = wl.c2()
fit = e.gp_fit(c2["x"], c2["y"], c2["h"], c2["w"], c2["s"])
for M in (256, 1000):
    xo = np.linspace(-5.0, 5.0, M) + 1e-3
    dev, cls, wall = bench._prof_call(e, lambda: fit.predict(xo), reps=5)
    print("M=%d kernels %.4f ms wall %.4f ms classes %s" % (M, dev, wall, {k: round(v, 4) for k, v in cls.items()}))
fit.close()
e.close()
