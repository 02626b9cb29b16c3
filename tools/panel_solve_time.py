"""ms and TFLOP/s (m kb^2 per matrix) of the batched panel solve at the shapes of a C5 shard's
outer blocks (64 x m2 x 448), both forms."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402

e = Engine(0)
rs = np.random.RandomState(0)
shapes = [(1472, 448, 64), (1024, 448, 64), (576, 448, 64), (128, 448, 64), (320, 256, 64),
          (3712, 448, 100), (896, 448, 256)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for (m, kb, batch) in shapes:
    G = rs.standard_normal((kb, kb))
    Lf = np.linalg.cholesky(G @ G.T + kb * np.eye(kb))
    Ls = np.repeat(Lf[None], batch, 0)
    Xs = rs.standard_normal((batch, m, kb))
    out = []
    for mode in (1, 2):
        ms = e.probe_panel_solve(Ls, Xs, mode, reps=10)
        out.append("mode %d %.3f ms %.1f TFLOP/s" % (mode, ms, m * kb * kb * batch / ms / 1e9))
    print("m %d kb %d batch %d: %s" % (m, kb, batch, "   ".join(out)), flush=True)
e.close()
