"""The tall-tile panel solve (trsm_sweep_tall_kernel<RT>): every RT against scipy and against the
64 x 64 tile's bits on ragged row counts, then ms / TFLOP/s per RT at the batched configs' shapes
(mode 2 = round 4's kernel, 3 = the launcher's pick, 10 + RT forced)."""
import os
import sys

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402

e = Engine(0)
rs = np.random.RandomState(0)
what = sys.argv[1] if len(sys.argv) > 1 else "all"
RTS = [4, 5, 6, 7, 8, 9]
if what in ("all", "check"):
    for (m, kb, batch) in [(64, 64, 1), (192, 128, 3), (320, 448, 9), (576, 192, 2), (1472, 448, 3), (256, 256, 10), (128, 320, 2), (64, 384, 1)]:
        Ls, Xs, refs = [], [], []
        for b in range(batch):
            G = rs.standard_normal((kb, kb))
            Lf = np.linalg.cholesky(G @ G.T + kb * np.eye(kb))
            X = rs.standard_normal((m, kb))
            Ls.append(Lf)
            Xs.append(X)
            refs.append(sla.solve_triangular(Lf, X.T, lower=True).T)
        Ls, Xs, refs = np.array(Ls), np.array(Xs), np.array(refs)
        old = e.probe_panel_solve(Ls, Xs, 2)
        line = []
        for rt in RTS:
            out = e.probe_panel_solve(Ls, Xs, 10 + rt)
            err = np.abs(out - refs).max() / np.abs(refs).max()
            line.append("rt%d %.1e%s" % (rt, err, "=" if np.array_equal(out, old) else "!"))
        if kb <= 448:
            out = e.probe_panel_solve(Ls, Xs, 4)
            err = np.abs(out - refs).max() / np.abs(refs).max()
            line.append("rl %.1e%s" % (err, "=" if np.array_equal(out, old) else "!"))
        print("m %d kb %d batch %d: old %.1e  %s" % (
            m, kb, batch, np.abs(old - refs).max() / np.abs(refs).max(), " ".join(line)), flush=True)
if what in ("all", "time"):
    shapes = [(1472, 448, 64), (1024, 448, 64), (576, 448, 64), (128, 448, 64), (3712, 448, 100),
              (3328, 384, 100), (896, 448, 256), (640, 384, 256)]
    for (m, kb, batch) in shapes:
        G = rs.standard_normal((kb, kb))
        Lf = np.linalg.cholesky(G @ G.T + kb * np.eye(kb))
        Ls = np.repeat(Lf[None], batch, 0)
        Xs = rs.standard_normal((batch, m, kb))
        out = []
        for mode in [2, 3] + [10 + rt for rt in RTS]:
            ms = e.probe_panel_solve(Ls, Xs, mode, reps=10)
            out.append("%s %.3f/%.1f" % ({2: "old", 3: "auto"}.get(mode, "rt%d" % (mode - 10)), ms,
                                         m * kb * kb * batch / ms / 1e9))
        print("m %d kb %d batch %d (ms/TFLOPs): %s" % (m, kb, batch, "  ".join(out)), flush=True)
e.close()
