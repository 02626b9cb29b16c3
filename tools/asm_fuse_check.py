"""The fused assembly (BQ_ASM_FUSE, default on: a batched plan assembles only its first outer
block's columns, block 0's products compute the rest of the system instead of loading it) against
the full assembly: the same bits on dense 2-D batches, a C5 shard, 256 x C2 and a C3 chunk, and the
time of each."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402


def engine(fuse):
    os.environ["BQ_ASM_FUSE"] = "1" if fuse else "0"
    try:
        return Engine(0)
    finally:
        del os.environ["BQ_ASM_FUSE"]


def run_plan(e, B, d, n, M, x, y, xo, h, w, s, reps=6):
    plan = e.plan(B, d, n, M)
    plan.set_inputs(x, y, xo, h, w, s)
    ts = []
    for _ in range(reps):
        e.sync()
        e.timer_start()
        plan.run()
        ts.append(e.timer_stop_ms())
    res = plan.results()
    plan.close()
    return res, min(ts[1:])


def dense(batch, n, m, d=2, seed=11):
    rs = np.random.RandomState(seed)
    x = rs.uniform(-3, 3, (batch, d, n))
    xo = rs.uniform(-3, 3, (batch, d, m))
    y = wl.norm_logpdf(x[:, 0]) + wl.norm_logpdf(x[:, 1])
    return x, y, xo, 1.3, np.full(d, 6.0 / np.sqrt(n) * 1.5), 0.05


e0, e1 = engine(False), engine(True)
cases = []
for (batch, n, m) in [(12, 1100, 70), (100, 700, 40), (64, 2048, 40), (96, 1100, 40), (5, 3000, 33)]:
    x, y, xo, h, w, s = dense(batch, n, m)
    cases.append(("dense %d x (%d, %d)" % (batch, n, m), batch, 2, n, m, x, y, xo, h, w, s))
c5 = wl.c5(range(64))
cases.append(("C5 shard", 64, 1, 2048, 256, c5["x"], c5["y"], c5["xo"], c5["h"], c5["w"], c5["s"]))
c2 = wl.c2()
cases.append(("256 x C2", 256, 1, 1024, 256, np.repeat(c2["x"][None], 256, 0), np.repeat(c2["y"][None], 256, 0),
              np.repeat(c2["xo"][None], 256, 0), c2["h"], c2["w"], c2["s"]))
bad = 0
for (name, B, d, n, M, x, y, xo, h, w, s) in cases:
    r0, t0 = run_plan(e0, B, d, n, M, x, y, xo, h, w, s)
    r1, t1 = run_plan(e1, B, d, n, M, x, y, xo, h, w, s)
    same = all(np.array_equal(a, b) for a, b in zip(r0, r1))
    bad += not same
    print("%-24s full assembly %.3f ms   fused %.3f ms   same bits: %s   failed %d" % (
        name, t0, t1, same, int((r1[3] != 0).sum())), flush=True)
c3 = wl.c3()
lm = []
for e in (e0, e1):
    e.logml_grid(c3["x"], c3["y"], c3["h"][:100], c3["w"][:100], c3["s"], chunk=100)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        v = e.logml_grid(c3["x"], c3["y"], c3["h"], c3["w"], c3["s"], chunk=100)
        best = min(best, (time.perf_counter() - t) * 1e3)
    lm.append((v, best))
same = np.array_equal(lm[0][0], lm[1][0])
bad += not same
print("%-24s full assembly %.1f ms   fused %.1f ms   same bits: %s" % ("C3 grid 400 x 4096", lm[0][1], lm[1][1], same))
e0.close()
e1.close()
sys.exit(1 if bad else 0)
