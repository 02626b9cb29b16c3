"""potrf time of the C4-style system at several sizes, look-ahead on and off:
    python tools/potrf_sizes.py 2048 4096 6144 8192"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine, _lib as L_  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

e = Engine(0)
lib, ctx = e._lib, e._ctx
if os.environ.get("TB_NB"):
    e.set_block(int(os.environ["TB_NB"]))
for n in [int(a) for a in sys.argv[1:]]:
    c4 = wl.c4(n)
    w4 = np.ascontiguousarray(c4["w"])
    xd, Kd, info = e.alloc(8 * n), e.alloc(8 * n * n), e.alloc(64)
    e.upload(xd, np.ascontiguousarray(c4["x"]))
    out = []
    for la in (True, False):
        e.set_lookahead(la)
        best = 1e9
        for rep in range(4):
            e._check(lib.bq_gram_gauss_dev(ctx, xd, 1, n, c4["h"], L_.dptr(w4), c4["s"], Kd, n))
            e.sync()
            e.timer_start()
            e._check(lib.bq_potrf_dev(ctx, Kd, n, n, info))
            best = min(best, e.timer_stop_ms())
        out.append("%s %.3f ms (%.1f TF/s)" % ("la" if la else "seq", best, n ** 3 / 3.0 / best / 1e9))
    print(n, " | ".join(out), flush=True)
    e.free(xd), e.free(Kd), e.free(info)
e.close()
