"""Times the C3 hyper-parameter grid (400 points, N=4096, chunks of 100) several times."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

e = Engine(0)
c3 = wl.c3()
pre = os.environ.get("C3_PRE", "")
if "la" in pre:
    e.set_lookahead(True)
if "prof" in pre:
    e.profile(True)
    e.profile_reset()
    e.profile(False)
if "big" in pre:
    from bayesian_quadrature_amd import _lib as L_
    n = 16384
    c4 = wl.c4(n)
    w4 = np.ascontiguousarray(c4["w"])
    xd, Kd, info = e.alloc(8 * n), e.alloc(8 * n * n), e.alloc(64)
    e.upload(xd, np.ascontiguousarray(c4["x"]))
    e._check(e._lib.bq_gram_gauss_dev(e._ctx, xd, 1, n, c4["h"], L_.dptr(w4), c4["s"], Kd, n))
    e._check(e._lib.bq_potrf_dev(e._ctx, Kd, n, n, info))
    e.sync()
    e.free(info), e.free(xd), e.free(Kd)
if "c5" in pre:
    c5 = wl.c5(range(64))
    e.batch_fit_predict(c5["x"], c5["y"], c5["h"], c5["w"], c5["s"], c5["xo"])
chunk = int(os.environ.get("C3_CHUNK", "100"))
for rep in range(3):
    t0 = time.perf_counter()
    lm = e.logml_grid(c3["x"], c3["y"], c3["h"], c3["w"], c3["s"], chunk=chunk)
    print("rep", rep, "%.1f ms" % ((time.perf_counter() - t0) * 1e3), "n_inf", int(np.isinf(lm).sum()), flush=True)
e.close()
