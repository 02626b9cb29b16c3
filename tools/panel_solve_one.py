"""One shape of the batched panel solve, one mode, `reps` launches (a target for rocprofv3):
python tools/panel_solve_one.py MODE M KB BATCH [REPS]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402

mode, m, kb, batch = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
e = Engine(0)
rs = np.random.RandomState(0)
G = rs.standard_normal((kb, kb))
Lf = np.linalg.cholesky(G @ G.T + kb * np.eye(kb))
Ls = np.repeat(Lf[None], batch, 0)
Xs = rs.standard_normal((batch, m, kb))
ms = e.probe_panel_solve(Ls, Xs, mode, reps=reps)
print("mode %d m %d kb %d batch %d: %.3f ms %.1f TFLOP/s" % (mode, m, kb, batch, ms,
                                                             m * kb * kb * batch / ms / 1e9))
e.close()
