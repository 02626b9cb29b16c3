"""The batched panel solve against scipy on dense random factors: max relative error per 64-column
slab, modes 1 (recursive) and 2 (one-launch sweep)."""
import os
import sys

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402

e = Engine(0)
rs = np.random.RandomState(0)
for (m, kb, batch) in [(64, 64, 1), (64, 128, 1), (128, 192, 2), (256, 448, 3), (3712, 448, 2)]:
    Ls, Xs, refs = [], [], []
    for b in range(batch):
        G = rs.standard_normal((kb, kb))
        Lf = np.linalg.cholesky(G @ G.T + kb * np.eye(kb))
        X = rs.standard_normal((m, kb))
        Ls.append(Lf)
        Xs.append(X)
        refs.append(sla.solve_triangular(Lf, X.T, lower=True).T)
    Ls, Xs, refs = np.array(Ls), np.array(Xs), np.array(refs)
    for mode in (1, 2):
        out = e.probe_panel_solve(Ls, Xs, mode)
        err = np.abs(out - refs) / np.max(np.abs(refs))
        per = [float(err[:, :, 64 * s:64 * s + 64].max()) for s in range(kb // 64)]
        rows = [float(err[:, 64 * r:64 * r + 64, :].max()) for r in range(min(m // 64, 4))]
        print("m %d kb %d batch %d mode %d: max %.2e per slab %s first row blocks %s" % (
            m, kb, batch, mode, err.max(), " ".join("%.1e" % v for v in per),
            " ".join("%.1e" % v for v in rows)), flush=True)
e.close()
