import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
from bayesian_quadrature_amd import Engine, workloads as wl
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
e = Engine(0)
c = wl.c4(n)
fit = e.gp_fit(c["x"], wl.norm_logpdf(c["x"]), c["h"], c["w"], c["s"])
b = np.random.RandomState(n).randn(n)
fit.solve(b)
os.environ["X"] = "1"
rows = e.timeline(lambda: fit.solve(b))
print("launches", len(rows), "span %.3f ms" % max(r[3] for r in rows))
for cls, st, t0, t1, w in rows:
    print("%d %-14s %8.3f %8.1f us" % (st, cls, t0, (t1 - t0) * 1e3))
fit.close(); e.close()
