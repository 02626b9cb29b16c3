"""A/B of the batched sweep's switches on the same inputs: logml of a batch through a plan with
BQ_DF_SWEEP=1 against BQ_DF_SWEEP=0 (and the recursive panels, BQ_DIAG_FIRST=0), several sizes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402


def run(env, n, m, batch, d):
    for k, v in env.items():
        os.environ[k] = v
    e = Engine(0)
    rs = np.random.RandomState(5)
    if d == 1:
        x = np.sort(rs.uniform(-5, 5, (batch, n)), axis=1)
        xo = np.tile(np.linspace(-5, 5, max(m, 1))[None], (batch, 1))[:, :m]
        y = wl.norm_logpdf(x)
        w = np.array([10.0 / n])
    else:
        x = rs.uniform(-5, 5, (batch, d, n))
        xo = rs.uniform(-5, 5, (batch, d, m))
        y = wl.norm_logpdf(x[:, 0]) + wl.norm_logpdf(x[:, 1])
        w = np.full(d, 10.0 / np.sqrt(n))
    plan = e.plan(batch, d, n, m)
    plan.set_inputs(x, y, xo, 1.0, w, 0.1)
    out = []
    for _ in range(2):
        plan.run()
        res = plan.results()
        out.append(np.array(res[2]))
    plan.close()
    e.close()
    for k in env:
        del os.environ[k]
    return out


for (n, m, batch, d) in [(1024, 256, 16, 1), (2048, 256, 16, 1), (3072, 0, 8, 1), (4096, 0, 8, 1),
                         (4096, 0, 8, 2), (4096, 256, 8, 1), (2048, 2112, 8, 1)]:
    ref = run({"BQ_DIAG_FIRST": "0"}, n, m, batch, d)
    a = run({"BQ_DF_SWEEP": "0"}, n, m, batch, d)
    b = run({"BQ_DF_SWEEP": "1"}, n, m, batch, d)
    rel = lambda u, v: float(np.max(np.abs(u - v) / np.abs(v)))
    print("n %d m %d batch %d d %d: rec-vs-old %.2e sweep-vs-old %.2e %.2e repeat %.2e" % (
        n, m, batch, d, rel(a[0], ref[0]), rel(b[0], ref[0]), rel(b[1], ref[0]), rel(b[0], b[1])),
        flush=True)
