"""Launch timeline of one pass of a resident plan (eager launches, HIP events around every launch
on its own stream): python tools/plan_timeline.py c5|c2x256|c3chunk [v]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "c5"
e = Engine(0)
if what == "c5":
    c5 = wl.c5(range(64))
    plan = e.plan(64, 1, 2048, 256)
    plan.set_inputs(c5["x"], c5["y"], c5["xo"], c5["h"], c5["w"], c5["s"])
elif what == "c2x256":
    c2 = wl.c2()
    B = 256
    plan = e.plan(B, 1, 1024, 256)
    plan.set_inputs(np.repeat(c2["x"][None], B, 0), np.repeat(c2["y"][None], B, 0),
                    np.repeat(c2["xo"][None], B, 0), c2["h"], c2["w"], c2["s"])
else:
    c3 = wl.c3()
    B = 100
    plan = e.plan(B, 2, 4096, 0)
    plan.set_inputs(np.repeat(c3["x"][None], B, 0), np.repeat(c3["y"][None], B, 0),
                    np.zeros((B, 2, 0)), c3["h"][:B], c3["w"][:B], c3["s"])
for _ in range(3):
    plan.run()
e.sync()
rows = e.timeline(plan.run)
end = max(r[3] for r in rows)
print(what, "launches", len(rows), "span %.3f ms" % end)
by = {}
for cls, st, t0, t1, w in rows:
    by[cls] = by.get(cls, 0.0) + (t1 - t0)
print("class ms:", " ".join("%s %.3f" % kv for kv in sorted(by.items(), key=lambda kv: -kv[1])))
if len(sys.argv) > 2:
    for cls, st, t0, t1, w in rows:
        print("%d %-20s %8.3f %8.3f  %7.1f us" % (st, cls, t0, t1, (t1 - t0) * 1e3))
plan.close()
e.close()
