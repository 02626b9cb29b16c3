"""N = 16384 potrf over outer block and look-ahead hand-over threshold:
    python tools/potrf_sweep.py 256,320,384,512 2048,3072,4096,6144"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine, _lib as L_  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

e = Engine(0)
lib, ctx = e._lib, e._ctx
n = 16384
nbs = [int(a) for a in sys.argv[1].split(",")]
lms = [int(a) for a in sys.argv[2].split(",")]
c4 = wl.c4(n)
w4 = np.ascontiguousarray(c4["w"])
xd, Kd, info = e.alloc(8 * n), e.alloc(8 * n * n), e.alloc(64)
e.upload(xd, np.ascontiguousarray(c4["x"]))
for nb in nbs:
    e.set_block(nb)
    row = []
    for lm in lms:
        e.set_lookahead(True, lm)
        best = 1e9
        for rep in range(3):
            e._check(lib.bq_gram_gauss_dev(ctx, xd, 1, n, c4["h"], L_.dptr(w4), c4["s"], Kd, n))
            e.sync()
            e.timer_start()
            e._check(lib.bq_potrf_dev(ctx, Kd, n, n, info))
            best = min(best, e.timer_stop_ms())
        row.append("%d:%.2f" % (lm, best))
    print("nb", nb, " ".join(row), flush=True)
e.close()
