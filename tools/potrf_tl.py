"""Launch timeline of one N = 16384 (argv[1]) factorisation as shipped (engine block, look-ahead):
the engine's launch profiler; prints the launches after `argv[2]` ms (default: the last 3 ms)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine, _lib as L_, workloads as wl  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
e = Engine(0)
if os.environ.get("TL_NB"):
    e.set_block(int(os.environ["TL_NB"]))
if os.environ.get("TL_LA_MIN"):
    e.set_lookahead(True, int(os.environ["TL_LA_MIN"]))
c4 = wl.c4(n)
w4 = np.ascontiguousarray(c4["w"])
xd, Kd, info = e.alloc(8 * n), e.alloc(8 * n * n), e.alloc(64)
e.upload(xd, np.ascontiguousarray(c4["x"]))


def gram():
    e._check(e._lib.bq_gram_gauss_dev(e._ctx, xd, 1, n, c4["h"], L_.dptr(w4), c4["s"], Kd, n))


def potrf():
    e._check(e._lib.bq_potrf_dev(e._ctx, Kd, n, n, info))


for rep in range(3):
    gram()
    e.sync()
    e.timer_start()
    potrf()
    print("potrf ms %.3f" % e.timer_stop_ms(), flush=True)
gram()
e.sync()
rows = e.timeline(potrf)
end = max(r[3] for r in rows)
print("launches", len(rows), "span %.3f ms (under the launch profiler)" % end)
t_from = float(sys.argv[2]) if len(sys.argv) > 2 else end - 3.0
for cls, st, t0, t1, w in rows:
    if t1 >= t_from:
        print("%d %-20s %8.3f %8.1f us %6.1f TFLOP/s" % (st, cls, t0, (t1 - t0) * 1e3,
                                                        w / max(t1 - t0, 1e-9) / 1e9))
e.close()
