"""Launch timeline of one C5 shard pass (eager launches, HIP events around every launch on its
own stream -- the engine's launch profiler): which kernels of the two streams run when."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

e = Engine(0)
c5 = wl.c5(range(64))
plan = e.plan(64, 1, 2048, 256)
plan.set_inputs(c5["x"], c5["y"], c5["xo"], c5["h"], c5["w"], c5["s"])
for _ in range(3):
    plan.run()
e.sync()
rows = e.timeline(plan.run)
end = max(r[3] for r in rows)
print("launches", len(rows), "span %.3f ms" % end)
busy = {0: 0.0, 1: 0.0}
for cls, st, t0, t1, w in rows:
    busy[st] += t1 - t0
print("stream busy ms", busy)
if len(sys.argv) > 1:
    for cls, st, t0, t1, w in rows:
        print("%d %-20s %8.3f %8.3f  %7.1f us" % (st, cls, t0, t1, (t1 - t0) * 1e3))
plan.close()
e.close()
