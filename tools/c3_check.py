"""C3 grid (or its first `chunk` points) through logml_grid, n_inf and a checksum: a quick A/B of
launch-sequence switches (BQ_DIAG_FIRST, BQ_DF_SWEEP, BQ_LOOKAHEAD ...) on the same inputs."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

e = Engine(0)
c3 = wl.c3()
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 400
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 100
for rep in range(2):
    t0 = time.perf_counter()
    lm = e.logml_grid(c3["x"], c3["y"], c3["h"][:npts], c3["w"][:npts], c3["s"], chunk=chunk)
    print("c3 %d pts chunk %d: %.1f ms n_inf %d sum %.12e" % (
        npts, chunk, (time.perf_counter() - t0) * 1e3, int(np.isinf(lm).sum()),
        float(lm[np.isfinite(lm)].sum())), flush=True)
e.close()
