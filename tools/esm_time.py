"""Acquisition loop at the plotting size of the reference (bq.py:773-774: 1000 candidates):
bq_esm_border (bordered update of the resident factor) against bq_esm_batch (one batched
refactorisation per candidate) at nsc = 1024."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

if __name__ == "__main__":
    e = Engine(0)
    ns, nc, M = 1000, 24, 1000
    rs = np.random.RandomState(1)
    xs = np.linspace(-5, 5, ns)
    dx = 10.0 / (ns - 1)
    xc = np.sort(rs.uniform(-6, 6, nc))
    xc = xc[np.min(np.abs(xc[:, None] - xs[None]), axis=1) > 0.3 * dx][:nc]
    x_sc = np.concatenate([xs, xc])
    l_sc = np.exp(wl.norm_logpdf(x_sc))
    x_a = np.linspace(-7, 7, M) + 1e-3
    mu, cov = np.array([0.0]), np.array([[10.0]])
    h, w, thresh = 0.2, 1.04 * dx, 0.5
    fit = e.gp_fit(x_sc, l_sc, h, w, 0.0)
    out = {"ns": ns, "nc": int(len(xc)), "M": M}
    for name, fn in (("esm_border", lambda: e.esm_border(fit, ns, x_a, thresh, mu, cov)),
                     ("esm_batch", lambda: e.esm_batch(x_sc, l_sc, ns, x_a, h, w, thresh, mu, cov))):
        r = fn()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            r = fn()
        out[name + "_ms"] = (time.perf_counter() - t0) / reps * 1e3
        out[name + "_failed"] = int((r[2] != 0).sum())
        out[name] = r
    a, b = out.pop("esm_border"), out.pop("esm_batch")
    sc = max(np.abs(b[0]).max(), np.abs(b[1]).max())
    out["max_rel_diff"] = float(max(np.abs(a[0] - b[0]).max(), np.abs(a[1] - b[1]).max()) / sc)
    out["speedup"] = out["esm_batch_ms"] / out["esm_border_ms"]
    print(json.dumps(out, indent=1))
